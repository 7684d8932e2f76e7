// cmvm_core.h -- scalar building blocks of the CMVM greedy engine, shared verbatim by the HIP kernels
// (cmvm_engine.hip) and by the sequential reformulation model used in tests (tests/model/engine_model.cc).
//
// Everything here is integer / bit manipulation plus a handful of IEEE-exact float operations, written so
// that host (g++) and device (hipcc --offload-arch=gfx950 -ffp-contract=off) produce identical bits.
//
// Data model (differs from the reference's sparse int8 digit lists, types.hh:104-141):
//   * a CELL holds all CSD digits of one (row, output column): two position bitmasks, `plus` in the low
//     half and `minus` in the high half of a 32-bit (n_bits <= 16) or 64-bit (n_bits <= 32) word;
//   * a pair KEY index enumerates (sub, shift): idx = sub * (2N-1) + shift + (N-1), so that a larger idx
//     is exactly a larger (sub, shift) in the reference's Pair ordering (types.hh:28-36);
//   * a BLOCK holds the occurrence counts of all K = 2(2N-1) keys of one row pair (id0 <= id1).
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DA_HD __host__ __device__ __forceinline__
#else
#define DA_HD inline
#endif

namespace da {

enum Method : int { M_MC = 0, M_MC_DC = 1, M_MC_PDC = 2, M_WMC = 3, M_WMC_DC = 4, M_WMC_PDC = 5, M_DUMMY = 6 };

enum ChainError : int {
    E_OK = 0,
    E_ROW_CAPACITY = 1,    // more greedy iterations than the row arena holds
    E_TABLE_CAPACITY = 2,  // pair-block table arena too small
    E_LIST_CAPACITY = 3,   // a column row-list overflowed (cannot happen with the exact bound; guarded anyway)
    E_FLOAT_DOMAIN = 4,    // the latency model met a step it has no logarithm for (subnormal / non-positive)
    E_COUNT_OVERFLOW = 5,  // a pair count exceeded 16 bits
    E_REMOTE_ERROR = 6,    // column-sharded chain: another rank failed with something other than a capacity error, or the ranks fell out of step -- not worth a retry
};

struct RowInfo {  // quantised interval + latency of one row (reference Op.qint / Op.latency)
    float lo, hi, step, lat;
};

// ----------------------------------------------------------------------------------------------- bits
DA_HD uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
DA_HD float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
DA_HD int popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}
DA_HD int ctz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)x) - 1;
#else
    return __builtin_ctz(x);
#endif
}

template <class Cell> struct CellOps;
template <> struct CellOps<uint32_t> {
    static constexpr int HALF = 16;
    static DA_HD uint32_t plus(uint32_t c) { return c & 0xFFFFu; }
    static DA_HD uint32_t minus(uint32_t c) { return c >> 16; }
    static DA_HD uint32_t make(uint32_t p, uint32_t m) { return p | (m << 16); }
};
template <> struct CellOps<uint64_t> {
    static constexpr int HALF = 32;
    static DA_HD uint32_t plus(uint64_t c) { return (uint32_t)c; }
    static DA_HD uint32_t minus(uint64_t c) { return (uint32_t)(c >> 32); }
    static DA_HD uint64_t make(uint32_t p, uint32_t m) { return (uint64_t)p | ((uint64_t)m << 32); }
};

DA_HD int key_count(int n_bits) { return 2 * (2 * n_bits - 1); }
DA_HD int key_index(int shift, int sub, int n_bits) { return sub * (2 * n_bits - 1) + shift + (n_bits - 1); }
DA_HD void key_decode(int idx, int n_bits, int &shift, int &sub) {
    int w = 2 * n_bits - 1;
    sub = idx >= w;
    shift = idx - sub * w - (n_bits - 1);
}

// bit p of the result = bit (p + s) of x   (s may be negative)
DA_HD uint32_t take_from(uint32_t x, int s) { return s >= 0 ? (s >= 32 ? 0u : x >> s) : (-s >= 32 ? 0u : x << (-s)); }
// bit (p + s) of the result = bit p of x
DA_HD uint32_t move_by(uint32_t x, int s) { return take_from(x, -s); }

// ------------------------------------------------------------------------------------ CSD (NAF) digits
// Non-adjacent form of x: identical digits to the reference's threshold recoding (bit_decompose.cc:22-42);
// checked exhaustively against the oracle in tests/test_oracle.py.
DA_HD void naf_masks(int32_t x, uint32_t &plus, uint32_t &minus) {
    uint32_t a = x < 0 ? (uint32_t)(-(int64_t)x) : (uint32_t)x;
    uint64_t x3 = (uint64_t)a * 3u;
    uint32_t pos = (uint32_t)((x3 & ~(uint64_t)a) >> 1);
    uint32_t neg = (uint32_t)((~x3 & (uint64_t)a) >> 1);
    plus = x < 0 ? neg : pos;
    minus = x < 0 ? pos : neg;
}
DA_HD int naf_weight(int32_t x) {
    uint32_t a = x < 0 ? (uint32_t)(-(int64_t)x) : (uint32_t)x;
    uint64_t x3 = (uint64_t)a * 3u;
    uint64_t d = (x3 ^ (uint64_t)a) >> 1;
    return popc32((uint32_t)d) + popc32((uint32_t)(d >> 32));
}
// digit width N = max(1, ceil(log2(max(maxabs,1) * 1.5)))  (bit_decompose.cc:24-27), in exact integers:
// the smallest N >= 1 with 2^(N+1) >= 3 * maxabs (3*maxabs is never a power of two, so no rounding case).
DA_HD int csd_width(uint32_t max_abs) {
    uint64_t t = 3ull * (max_abs ? max_abs : 1u);
    int n = 1;
    while ((1ull << (n + 1)) < t) ++n;
    return n;
}
// lowest set bit position of a float (bit_decompose.cc:10-20); 127 for zero
DA_HD int lsb_loc(float x) {
    if (x == 0.0f) return 127;
    uint32_t b = f2u(x);
    int e = (int)((b >> 23) & 0xFF);
    uint32_t m = (b & 0x7FFFFFu) + (1u << 23);
    return (int)(int8_t)(e + ctz32(m) - 150);
}
// ceil(log2 x) from the float bit pattern (indexers.hh:12-18), including its int8 wrap
DA_HD int iceil_log2(float x) {
    uint32_t b = f2u(x);
    int e = (int)((b >> 23) & 0xFF);
    return (int)(int8_t)(e - 127 + ((b & 0x7FFFFFu) != 0));
}
// exact 2^s as float for |s| <= 126
DA_HD float pow2f(int s) { return u2f((uint32_t)(s + 127) << 23); }

// ------------------------------------------------------------------------------------ pair occurrences
// Visit every digit pair between cell `lo` (row with the smaller id) and cell `hi` of one column and call
// f(key_index, +1).  Cross-row rule: shift = pos(hi digit) - pos(lo digit), sub = signs differ
// (state_opr.cc:69-77,133-137).
template <class Cell, class F> DA_HD void for_pairs_cross(Cell lo, Cell hi, int n_bits, F &&f) {
    using O = CellOps<Cell>;
    uint32_t lp = O::plus(lo), lm = O::minus(lo), hp = O::plus(hi), hm = O::minus(hi);
    uint32_t la = lp | lm, ha = hp | hm;
    while (la) {
        int p = ctz32(la);
        la &= la - 1;
        int sl = (lm >> p) & 1;
        uint32_t h = ha;
        while (h) {
            int q = ctz32(h);
            h &= h - 1;
            int sh = (hm >> q) & 1;
            f(key_index(q - p, sl ^ sh, n_bits));
        }
    }
}
// Same-row rule: ordered (higher digit, lower digit) => shift = pos(lower) - pos(higher) < 0
// (state_opr.cc:124-131).
template <class Cell, class F> DA_HD void for_pairs_self(Cell c, int n_bits, F &&f) {
    using O = CellOps<Cell>;
    uint32_t cp = O::plus(c), cm = O::minus(c);
    uint32_t all = cp | cm;
    uint32_t hi = all;
    while (hi) {
        int a = ctz32(hi);
        hi &= hi - 1;
        int sa = (cm >> a) & 1;
        uint32_t lo = all & ((1u << a) - 1u);
        while (lo) {
            int b = ctz32(lo);
            lo &= lo - 1;
            int sb = (cm >> b) & 1;
            f(key_index(b - a, sa ^ sb, n_bits));
        }
    }
}
// pairs between a SUBSET of digits `part` of the row `x` and all digits of the other row cell `y`;
// x_is_lo tells which of the two rows has the smaller id.
template <class Cell, class F> DA_HD void for_pairs_part(Cell part, Cell y, bool x_is_lo, int n_bits, F &&f) {
    if (x_is_lo)
        for_pairs_cross<Cell>(part, y, n_bits, f);
    else
        for_pairs_cross<Cell>(y, part, n_bits, f);
}

// ------------------------------------------------------------------------------------ substitution
// Apply the chosen pair (id0=A, id1=B, shift, sub) to one column (state_opr.cc:249-280).
// a, b: cells of rows A and B (b ignored when same_row).  Outputs the digits consumed from A (ma), from B
// (mb) and the cell of the new row (= the consumed A-role digits, position and sign preserved).
template <class Cell> DA_HD void substitute_column(Cell a, Cell b, bool same_row, int shift, int sub, Cell &ma, Cell &mb) {
    using O = CellOps<Cell>;
    uint32_t ap = O::plus(a), am = O::minus(a);
    if (!same_row) {
        uint32_t bp = O::plus(b), bm = O::minus(b);
        uint32_t map = ap & take_from(sub ? bm : bp, shift);
        uint32_t mam = am & take_from(sub ? bp : bm, shift);
        uint32_t tp = move_by(map, shift), tm = move_by(mam, shift);
        ma = O::make(map, mam);
        mb = sub ? O::make(tm, tp) : O::make(tp, tm);
        return;
    }
    // same row: shift < 0; walk the lower digit position upward, each digit used at most once
    int rel = -shift;
    uint32_t avp = ap, avm = am, map = 0, mam = 0, mbp = 0, mbm = 0;
    for (int p = 0; p + rel < O::HALF; ++p) {
        uint32_t bit_lo = 1u << p, bit_hi = 1u << (p + rel);
        uint32_t all = avp | avm;
        if (!(all & bit_lo) || !(all & bit_hi)) continue;
        int s_lo = (avm & bit_lo) != 0, s_hi = (avm & bit_hi) != 0;
        if ((s_lo ^ s_hi) != sub) continue;
        avp &= ~(bit_lo | bit_hi);
        avm &= ~(bit_lo | bit_hi);
        if (s_hi)
            mam |= bit_hi;
        else
            map |= bit_hi;
        if (s_lo)
            mbm |= bit_lo;
        else
            mbp |= bit_lo;
    }
    ma = O::make(map, mam);  // A-role digits (higher position) -> new row
    mb = O::make(mbp, mbm);  // B-role digits (lower position)
}

// ------------------------------------------------------------------------------------ scoring
// n_overlap of overlap_and_accum (indexers.cc:36-56), int8 arithmetic as in the reference
DA_HD int n_overlap(const RowInfo &a, const RowInfo &b) {
    float hi0 = a.hi + a.step, hi1 = b.hi + b.step;
    float st = a.step > b.step ? a.step : b.step;  // std::max(step0, step1)
    int8_t f = (int8_t)(-iceil_log2(st));
    float m0 = __builtin_fabsf(a.lo) < __builtin_fabsf(hi0) ? __builtin_fabsf(hi0) : __builtin_fabsf(a.lo);
    float m1 = __builtin_fabsf(b.lo) < __builtin_fabsf(hi1) ? __builtin_fabsf(hi1) : __builtin_fabsf(b.lo);
    int8_t i_low = (int8_t)iceil_log2(m1 < m0 ? m1 : m0);
    int8_t k = (a.lo < 0 || b.lo < 0) ? 1 : 0;
    return (int)(int8_t)(k + i_low + f);
}

// order-preserving map float -> u32 rank >= 1 (0 is reserved for "not selectable"); -0 == +0; NaN -> 0
DA_HD uint32_t float_rank(float s) {
    if (s != s) return 0;
    if (s == 0.0f) s = 0.0f;
    uint32_t u = f2u(s == 0.0f ? 0.0f : s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// rank of one table entry; 0 = cannot be selected.  Larger rank == larger reference score; equal rank ==
// equal score (indexers.cc:6-90).  `ov` = n_overlap, `dl` = |lat0 - lat1|.
DA_HD uint32_t entry_rank(uint32_t count, int ov, float dl, int method) {
    if (count < 2) return 0;
    switch (method) {
    case M_MC: return count + 1;
    case M_WMC: {
#if defined(__HIP_DEVICE_COMPILE__)
        // counts are stored as u16 and |ov| < 2^7: the product fits 32 bits (the 64-bit multiply was four vector
        // instructions, eight times per partner pass)
        if (count <= 0xFFFFu) {
            const int32_t s32 = (int32_t)count * ov;
            return s32 >= 0 ? (uint32_t)s32 + 1 : 0;
        }
#endif
        int64_t s = (int64_t)count * ov;
        return s >= 0 ? (uint32_t)s + 1 : 0;
    }
    case M_MC_DC:
    case M_MC_PDC: {
#if defined(__HIP_DEVICE_COMPILE__)
        float s = __fsub_rn((float)count, __fmul_rn(1e9f, dl));
#else
        float pen = 1e9f * dl;
        float s = (float)count - pen;
#endif
        if (method == M_MC_DC && !(s >= 0.0f)) return 0;
        return float_rank(s);
    }
    case M_WMC_DC:
    case M_WMC_PDC: {
        uint32_t prod = count * (uint32_t)ov;  // unsigned wrap for negative overlap, as in the reference
#if defined(__HIP_DEVICE_COMPILE__)
        float s = __fsub_rn((float)prod, __fmul_rn(256.0f, dl));
#else
        float pen = 256.0f * dl;
        float s = (float)prod - pen;
#endif
        if (method == M_WMC_DC && !(s >= 0.0f)) return 0;
        return float_rank(s);
    }
    default: return 0;
    }
}

// tie-break word: larger == later in the reference's sorted table (id1, id0, sub, shift)
DA_HD uint64_t tie_word(uint32_t id0, uint32_t id1, int idx) {
    return ((uint64_t)id1 << 31) | ((uint64_t)id0 << 7) | (uint64_t)idx;
}

// ------------------------------------------------------------------------------------ qint / latency
// qint_add(q0, q1, shift, false, sub) (state_opr.cc:8-29); all operations exact or single-rounded IEEE
DA_HD void qint_add_pair(const RowInfo &a, const RowInfo &b, int shift, int sub, float &lo, float &hi, float &step) {
    float lo1 = b.lo, hi1 = b.hi, st1 = b.step;
    if (sub) {
        float t = lo1;
        lo1 = -hi1;
        hi1 = -t;
    }
    float s = pow2f(shift);
#if defined(__HIP_DEVICE_COMPILE__)
    lo1 = __fmul_rn(lo1, s);
    hi1 = __fmul_rn(hi1, s);
    st1 = __fmul_rn(st1, s);
    lo = __fadd_rn(a.lo, lo1);
    hi = __fadd_rn(a.hi, hi1);
#else
    lo1 = lo1 * s;
    hi1 = hi1 * s;
    st1 = st1 * s;
    lo = a.lo + lo1;
    hi = a.hi + hi1;
#endif
    step = a.step < st1 ? a.step : st1;  // std::min(step0, step1)
}

// Host-measured description of glibc's log2f near powers of two: ceil(log2f(x)) for x = 2^e * (1 + t*2^-23)
// equals e (not e+1) iff t <= tie[e + 150].  Filled by the host at library start-up from the local libm, so the
// device latency model reproduces std::ceil(std::log2(float)) of this very machine bit for bit.
struct Log2Table {
    uint8_t tie[280];
};

// std::ceil(std::log2(x)) for float x >= 0 (cost_add, state_opr.cc:58-62)
DA_HD float ceil_log2f_emul(float x, const Log2Table &tab, int &domain_err) {
    uint32_t b = f2u(x);
    int e = (int)((b >> 23) & 0xFF);
    uint32_t m = b & 0x7FFFFFu;
    if (x != x) return x;
    if (x == 0.0f) return -__builtin_inff();
    if (e == 255) return __builtin_inff();
    if (e == 0) {  // subnormal bounds never occur for fixed-point intervals
        domain_err = 1;
        return 0.0f;
    }
    int ee = e - 127;
    if (m == 0 || m <= tab.tie[ee + 150]) return (float)ee;
    return (float)(ee + 1);
}
// -std::log2(step) (state_opr.cc:57).  A power of two is read off the exponent.  Any other step gets the HOST libm's value from a
// table: every row's step is an input step times a power of two (qint_add keeps min(step0, step1 * 2^shift)), so only the
// mantissas of the input steps ever occur; for each of them the host tabulates -log2f(mantissa * 2^(e - 127)) for all 254
// normal exponents with its own std::log2 -- exact by construction, no libm re-implementation on the device.  (One table row per
// distinct mantissa, as many as the inputs have: the look-up is a linear search, and matrices with more than a handful of
// different odd step mantissas do not come out of the tracer.)
struct StepLog2 {
    int n;                 // distinct non-power-of-two step mantissas of the chain's inputs
    const uint32_t *mant;  // [n] their 23 mantissa bits
    const float *tab;      // [n][256] -log2f of (mantissa, biased exponent)
};
DA_HD float neg_log2f_step(float st, const StepLog2 &sl, int &domain_err) {
    uint32_t b = f2u(st);
    int e = (int)((b >> 23) & 0xFF);
    if (st != st) return st;
    if (st == 0.0f) return __builtin_inff();
    if (e == 255 && (b & 0x7FFFFFu) == 0) return (b >> 31) ? st : -__builtin_inff();
    if ((b >> 31) || e == 0 || e == 255) {
        domain_err = 1;
        return 0.0f;
    }
    const uint32_t m = b & 0x7FFFFFu;
    if (m == 0) return -(float)(e - 127);
    for (int i = 0; i < sl.n; ++i)
        if (sl.mant[i] == m) return sl.tab[i * 256 + e];
    domain_err = 1;
    return 0.0f;
}
// latency increment of cost_add (state_opr.cc:31-67); the cost itself is recomputed on the host
DA_HD float adder_dlat(const RowInfo &a, const RowInfo &b, int shift, int sub, int adder_size, int carry_size,
                       const Log2Table &tab, const StepLog2 &sl, int &domain_err) {
    if (adder_size < 0 && carry_size < 0) return 1.0f;
    if (carry_size < 0) carry_size = 65535;
    float lo0 = a.lo, hi0 = a.hi, st0 = a.step, lo1 = b.lo, hi1 = b.hi, st1 = b.step;
    if (sub) {
        float t = lo1;
        lo1 = hi1;
        hi1 = t;
    }
    float sf = pow2f(shift);
#if defined(__HIP_DEVICE_COMPILE__)
    lo1 = __fmul_rn(lo1, sf);
    hi1 = __fmul_rn(hi1, sf);
    st1 = __fmul_rn(st1, sf);
    hi0 = __fadd_rn(hi0, st0);
    hi1 = __fadd_rn(hi1, st1);
#else
    lo1 = lo1 * sf;
    hi1 = hi1 * sf;
    st1 = st1 * sf;
    hi0 = hi0 + st0;
    hi1 = hi1 + st1;
#endif
    float f = neg_log2f_step(st0 > st1 ? st0 : (st1 > st0 ? st1 : st0), sl, domain_err);
    float m = __builtin_fabsf(lo0);
    float t1 = __builtin_fabsf(lo1), t2 = __builtin_fabsf(hi0), t3 = __builtin_fabsf(hi1);
    m = m < t1 ? t1 : m;
    m = m < t2 ? t2 : m;
    m = m < t3 ? t3 : m;
    float i = ceil_log2f_emul(m, tab, domain_err);
    int k = (a.lo < 0 || b.lo < 0) ? 1 : 0;
#if defined(__HIP_DEVICE_COMPILE__)
    float n_accum = __fadd_rn(__fadd_rn((float)k, i), f);
    return ceilf(__fdiv_rn(n_accum, (float)carry_size));
#else
    float n_accum = (float)k + i;
    n_accum = n_accum + f;
    float q = n_accum / (float)carry_size;
    return __builtin_ceilf(q);
#endif
}

}  // namespace da
