// cmvm_oracle.cc -- CPU ORACLE (test infrastructure, NOT the product path).
//
// A plain-C++ restatement of the reference CMVM optimiser (calad0i/da4ml,
// src/da4ml/_binary/cmvm/*.cc) used ONLY as the bit-exactness checker in tests/,
// in __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.  Nothing in
// da4ml_amd/ may import, link or call it.
//
// Pinning status: the reference's own tests hold no golden op lists (only
// `sol.kernel == kernel` property tests, tests/test_cmvm.py:23-55).  This
// restatement is pinned against the REAL reference sources compiled against a
// container shim (oracle/_ref, see oracle/Makefile and oracle/README.md) on
// seeded matrices; the golden fixtures in tests/golden/ were produced by that
// build (tests/golden/make_golden.py).
//
// The algorithmic structure (sorted pair table, purge + regenerate per
// iteration) deliberately follows the reference so that timing this file is a
// fair "port" CPU baseline.  Containers are std::vector; xtensor expressions of
// the reference are written out as loops.
//
// Every function cites the reference file:line it restates (paths relative to
// /root/reference/src/da4ml/_binary/cmvm/).

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

// ---------------------------------------------------------------- types.hh:10-37
struct QInt {
    float lo, hi, step;
};
struct OpRec {
    int64_t id0, id1, opcode, data;
    QInt q;
    float latency, cost;
};
struct PairKey {
    int64_t id0, id1;
    int8_t shift;
    bool sub;
    bool operator==(const PairKey &o) const {
        return id0 == o.id0 && id1 == o.id1 && shift == o.shift && sub == o.sub;
    }
    // ordering (id1, id0, sub, shift): types.hh:28-36
    bool operator<(const PairKey &o) const {
        return std::tie(id1, id0, sub, shift) < std::tie(o.id1, o.id0, o.sub, o.shift);
    }
};
using Entry = std::pair<PairKey, uint32_t>;

// digits are stored as +-(position+1), ascending position (types.hh:104-141)
static inline int dpos(int8_t v) { return std::abs((int)v) - 1; }
static inline int dsgn(int8_t v) { return v > 0 ? 1 : -1; }

struct Expr {  // one row of the digit tensor: cols[i_out] = digit list
    std::vector<std::vector<int8_t>> cols;
};

struct Stats {  // instrumentation for DESIGN.md / bench roofline (not in the reference)
    int64_t iterations = 0, p_init = 0, d0 = 0, f_first = 0, f_sum = 0, f_max = 0;
    int64_t live_sum = 0, match_sum = 0, regen_pairs = 0, tree_ops = 0;
};

struct State {  // types.hh:143-151
    std::vector<int8_t> shift0, shift1;
    std::vector<Expr> expr;
    int n_bits = 0;
    std::vector<OpRec> ops;
    std::vector<Entry> table;  // sorted, count >= 2 only
    int64_t n_in = 0, n_out = 0;
    Stats st;
};

struct Stage {  // types.hh:153-162
    int64_t n_in = 0, n_out = 0;
    std::vector<int64_t> inp_shifts, out_idxs, out_shifts, out_negs;
    std::vector<OpRec> ops;
    int carry_size = -1, adder_size = -1;
    Stats st;
};
struct Pipe {
    std::vector<Stage> stages;
};

// ---------------------------------------------------------------- bit_decompose.cc:10-20
static int8_t lsb_loc(float x) {
    if (x == 0.0f) return 127;
    uint32_t b;
    std::memcpy(&b, &x, 4);
    uint8_t e = (uint8_t)((b >> 23) & 0xFF);
    uint32_t m = b & 0x7FFFFFu;
    int tz = __builtin_ctz(m + (1u << 23));
    return (int8_t)(e + tz - 150);
}

// ---------------------------------------------------------------- indexers.hh:12-18
static int8_t iceil_log2(float x) {
    uint32_t b;
    std::memcpy(&b, &x, 4);
    uint8_t e = (uint8_t)((b >> 23) & 0xFF);
    uint32_t m = b & 0x7FFFFFu;
    return (int8_t)(e - 127 + (m != 0));
}

// ---------------------------------------------------------------- bit_decompose.cc:22-42
// Threshold recoding, digit n decided from the top; N from the GLOBAL max.
static int csd_width(int32_t max_abs) {
    double v = (double)std::max((float)max_abs, 1.0f) * 1.5;
    size_t n = (size_t)std::ceil(std::log2(v));
    return (int)std::max(n, (size_t)1);
}
static void csd_digits(int32_t x, int N, int8_t *out) {
    for (int n = N - 1; n >= 0; --n) {
        int32_t p = (int32_t)(1u << n);
        int32_t th = p * 2 / 3;
        int8_t d = (int8_t)((x > th) - (x < -th));
        out[n] = d;
        x -= p * (int32_t)d;
    }
}

// ---------------------------------------------------------------- bit_decompose.hh:25-34
// columns first (shift1), then rows (shift0); exact power-of-two scaling.
static void center(std::vector<float> &a, int64_t n_in, int64_t n_out, std::vector<int8_t> &s0,
                   std::vector<int8_t> &s1) {
    s0.assign(n_in, 0);
    s1.assign(n_out, 0);
    for (int64_t j = 0; j < n_out; ++j) {
        int8_t m = 127;
        for (int64_t i = 0; i < n_in; ++i) m = std::min(m, lsb_loc(a[i * n_out + j]));
        if (n_in == 0) m = 0;
        s1[j] = m;
    }
    for (int64_t i = 0; i < n_in; ++i)
        for (int64_t j = 0; j < n_out; ++j)
            a[i * n_out + j] = (float)((double)a[i * n_out + j] * std::pow(2.0, -(int)s1[j]));
    for (int64_t i = 0; i < n_in; ++i) {
        int8_t m = 127;
        for (int64_t j = 0; j < n_out; ++j) m = std::min(m, lsb_loc(a[i * n_out + j]));
        if (n_out == 0) m = 0;
        s0[i] = m;
    }
    for (int64_t i = 0; i < n_in; ++i)
        for (int64_t j = 0; j < n_out; ++j)
            a[i * n_out + j] = (float)((double)a[i * n_out + j] * std::pow(2.0, -(int)s0[i]));
}

// ---------------------------------------------------------------- bit_decompose.cc:45-62
static int csd_decompose(const float *kernel, int64_t n_in, int64_t n_out, bool do_center,
                         std::vector<int8_t> &csd, std::vector<int8_t> &s0, std::vector<int8_t> &s1) {
    std::vector<float> a(kernel, kernel + n_in * n_out);
    if (do_center)
        center(a, n_in, n_out, s0, s1);
    else {
        s0.assign(n_in, 0);
        s1.assign(n_out, 0);
    }
    std::vector<int32_t> xi(a.size());
    int32_t mx = 0;
    for (size_t k = 0; k < a.size(); ++k) {
        xi[k] = (int32_t)a[k];
        mx = std::max(mx, (int32_t)std::abs(xi[k]));
    }
    int N = csd_width(mx);
    csd.assign(a.size() * (size_t)N, 0);
    for (size_t k = 0; k < a.size(); ++k) csd_digits(xi[k], N, &csd[k * N]);
    return N;
}

// ---------------------------------------------------------------- state_opr.cc:8-29
static QInt qint_add(const QInt &a, const QInt &b, int64_t shift, bool neg_a, bool neg_b) {
    float lo0 = a.lo, hi0 = a.hi, st0 = a.step;
    float lo1 = b.lo, hi1 = b.hi, st1 = b.step;
    if (neg_a) {
        std::swap(lo0, hi0);
        lo0 = -lo0;
        hi0 = -hi0;
    }
    if (neg_b) {
        std::swap(lo1, hi1);
        lo1 = -lo1;
        hi1 = -hi1;
    }
    float s = (float)std::pow(2.0, (double)shift);
    lo1 *= s;
    hi1 *= s;
    st1 *= s;
    return QInt{lo0 + lo1, hi0 + hi1, std::min(st0, st1)};
}

// ---------------------------------------------------------------- state_opr.cc:31-67
static std::pair<float, float> cost_add(const QInt &a, const QInt &b, int64_t shift, bool sub, int adder_size,
                                        int carry_size) {
    if (adder_size < 0 && carry_size < 0) return {1.0f, 1.0f};
    if (adder_size < 0) adder_size = 65535;
    if (carry_size < 0) carry_size = 65535;
    float lo0 = a.lo, hi0 = a.hi, st0 = a.step;
    float lo1 = b.lo, hi1 = b.hi, st1 = b.step;
    if (sub) std::swap(lo1, hi1);
    float sf = (float)std::pow(2.0, (double)shift);
    lo1 *= sf;
    hi1 *= sf;
    st1 *= sf;
    hi0 += st0;
    hi1 += st1;
    float f = -std::log2(std::max(st0, st1));
    float i = std::ceil(std::log2(std::max({std::abs(lo0), std::abs(lo1), std::abs(hi0), std::abs(hi1)})));
    int k = (a.lo < 0 || b.lo < 0) ? 1 : 0;
    float n_accum = k + i + f;
    return {std::ceil(n_accum / carry_size), std::ceil(n_accum / adder_size)};
}

// ---------------------------------------------------------------- state_opr.cc:69-77
static inline PairKey make_pair_key(int64_t id0, int64_t id1, int8_t v0, int8_t v1) {
    if (id0 > id1) throw std::invalid_argument("id0 should be <= id1");
    return PairKey{id0, id1, (int8_t)(dpos(v1) - dpos(v0)), dsgn(v0) != dsgn(v1)};
}

// types.hh:73-95: sort raw pairs, run-length count, keep >= 2, merge into table
static void table_batch_add(std::vector<Entry> &table, std::vector<PairKey> &raw) {
    std::sort(raw.begin(), raw.end());
    std::vector<Entry> fresh;
    size_t n = raw.size();
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && raw[j] == raw[i]) ++j;
        if (j - i >= 2) fresh.emplace_back(raw[i], (uint32_t)(j - i));
        i = j;
    }
    std::vector<Entry> merged(table.size() + fresh.size());
    std::merge(table.begin(), table.end(), fresh.begin(), fresh.end(), merged.begin(),
               [](const Entry &x, const Entry &y) { return x.first < y.first; });
    table.swap(merged);
}

// pairs between two digit lists of one column (state_opr.cc:117-141 and :307-340)
static inline void emit_pairs(std::vector<PairKey> &raw, int64_t lo, int64_t hi, const std::vector<int8_t> &rlo,
                              const std::vector<int8_t> &rhi) {
    if (rlo.empty() || rhi.empty()) return;
    if (lo == hi) {
        for (size_t a = 1; a < rlo.size(); ++a)
            for (size_t b = 0; b < a; ++b) raw.push_back(make_pair_key(lo, lo, rlo[a], rlo[b]));
    } else {
        for (int8_t v0 : rlo)
            for (int8_t v1 : rhi) raw.push_back(make_pair_key(lo, hi, v0, v1));
    }
}

// ---------------------------------------------------------------- state_opr.cc:79-159
static State create_state(const float *kernel, int64_t n_in, int64_t n_out, const std::vector<QInt> &qints,
                          const std::vector<float> &lats, bool no_stat) {
    State s;
    s.n_in = n_in;
    s.n_out = n_out;
    std::vector<int8_t> csd;
    int N = csd_decompose(kernel, n_in, n_out, true, csd, s.shift0, s.shift1);
    for (int64_t i = 0; i < n_in; ++i)
        if (qints[i].lo == 0.0f && qints[i].hi == 0.0f)
            std::fill(csd.begin() + i * n_out * N, csd.begin() + (i + 1) * n_out * N, (int8_t)0);
    s.n_bits = N;
    s.expr.resize(n_in);
    for (int64_t i = 0; i < n_in; ++i) {
        s.expr[i].cols.resize(n_out);
        for (int64_t j = 0; j < n_out; ++j)
            for (int b = 0; b < N; ++b) {
                int8_t v = csd[(i * n_out + j) * N + b];
                if (v != 0) {
                    s.expr[i].cols[j].push_back((int8_t)(v * (b + 1)));
                    s.st.d0++;
                }
            }
    }
    if (!no_stat) {
        std::vector<PairKey> raw;
        for (int64_t j = 0; j < n_out; ++j)
            for (int64_t i0 = 0; i0 < n_in; ++i0)
                for (int64_t i1 = i0; i1 < n_in; ++i1) emit_pairs(raw, i0, i1, s.expr[i0].cols[j], s.expr[i1].cols[j]);
        s.st.p_init = (int64_t)raw.size();
        s.table.clear();
        table_batch_add(s.table, raw);
        s.st.f_first = (int64_t)s.table.size();
    }
    for (int64_t i = 0; i < n_in; ++i) s.ops.push_back(OpRec{i, -1, -1, 0, qints[i], lats[i], 0.0f});
    return s;
}

// ---------------------------------------------------------------- indexers.cc:6-90
static const PairKey NO_PAIR{-1, -1, 0, false};

static PairKey pick_mc(const State &s) {
    PairKey best = NO_PAIR;
    size_t top = 0;
    for (const auto &e : s.table)
        if (e.second >= top) {
            top = e.second;
            best = e.first;
        }
    return best;
}
static PairKey pick_mc_dc(const State &s, bool absolute) {
    PairKey best = NO_PAIR;
    float factor = 1e9f;
    float top = absolute ? 0.0f : -std::numeric_limits<float>::infinity();
    for (const auto &e : s.table) {
        float l0 = s.ops[e.first.id0].latency, l1 = s.ops[e.first.id1].latency;
        float score = e.second - factor * std::abs(l0 - l1);
        if (score >= top) {
            top = score;
            best = e.first;
        }
    }
    return best;
}
static std::pair<int8_t, int8_t> overlap_accum(const QInt &a, const QInt &b) {
    float lo0 = a.lo, hi0 = a.hi, st0 = a.step, lo1 = b.lo, hi1 = b.hi, st1 = b.step;
    hi0 += st0;
    hi1 += st1;
    int8_t f = (int8_t)(-iceil_log2(std::max(st0, st1)));
    int8_t i_high = iceil_log2((float)std::max({std::abs(lo0), std::abs(lo1), std::abs(hi0), std::abs(hi1)}));
    int8_t i_low =
        iceil_log2(std::min(std::max(std::abs(lo0), std::abs(hi0)), std::max(std::abs(lo1), std::abs(hi1))));
    int8_t k = (a.lo < 0 || b.lo < 0) ? 1 : 0;
    int8_t n_accum = (int8_t)(k + i_high + f);
    int8_t n_overlap = (int8_t)(k + i_low + f);
    return {n_overlap, n_accum};
}
static PairKey pick_wmc(const State &s) {
    int64_t top = 0;
    PairKey best = NO_PAIR;
    for (const auto &e : s.table) {
        int8_t ov = overlap_accum(s.ops[e.first.id0].q, s.ops[e.first.id1].q).first;
        int64_t score = (int64_t)e.second * ov;
        if (score >= top) {
            top = score;
            best = e.first;
        }
    }
    return best;
}
static PairKey pick_wmc_dc(const State &s, bool absolute) {
    float top = absolute ? 0.0f : -std::numeric_limits<float>::infinity();
    PairKey best = NO_PAIR;
    for (const auto &e : s.table) {
        int8_t ov = overlap_accum(s.ops[e.first.id0].q, s.ops[e.first.id1].q).first;
        float l0 = s.ops[e.first.id0].latency, l1 = s.ops[e.first.id1].latency;
        // uint32 * int8 -> unsigned arithmetic (wraps for negative overlap), then float (indexers.cc:83)
        float score = e.second * ov - 256 * std::abs(l0 - l1);
        if (score >= top) {
            top = score;
            best = e.first;
        }
    }
    return best;
}

// ---------------------------------------------------------------- state_opr.cc:211-283
static int find_pos(const std::vector<int8_t> &digits, int pos) {
    for (int i = 0; i < (int)digits.size(); ++i)
        if (dpos(digits[i]) == pos) return i;  // tombstones (0) have position -1
    return -1;
}
static void substitute(State &s, const PairKey &p, int adder_size, int carry_size) {
    const QInt &q0 = s.ops[p.id0].q, &q1 = s.ops[p.id1].q;
    auto [dlat, cost] = cost_add(q0, q1, p.shift, p.sub, adder_size, carry_size);
    float lat = std::max(s.ops[p.id0].latency, s.ops[p.id1].latency) + dlat;
    QInt nq = qint_add(q0, q1, p.shift, false, p.sub);
    s.ops.push_back(OpRec{p.id0, p.id1, (int64_t)p.sub, p.shift, nq, lat, cost});

    Expr fresh;
    fresh.cols.resize(s.n_out);
    int64_t ra = p.id0, rb = p.id1;
    int rel = p.shift;
    bool flip = false;
    if (rel < 0) {
        std::swap(ra, rb);
        rel = -rel;
        flip = true;
    }
    int want = p.sub ? -1 : 1;
    for (int64_t j = 0; j < s.n_out; ++j) {
        auto &da = s.expr[ra].cols[j];
        auto &db = s.expr[rb].cols[j];
        for (int ia = 0; ia < (int)da.size(); ++ia) {
            if (da[ia] == 0) continue;
            int pa = dpos(da[ia]), sa = dsgn(da[ia]);
            int pb = pa + rel;
            int ib = find_pos(db, pb);
            if (pb >= s.n_bits) continue;
            int sb = ib >= 0 ? dsgn(db[ib]) : 0;
            if (want * sb * sa != 1) continue;
            if (!flip)
                fresh.cols[j].push_back((int8_t)(sa * (pa + 1)));
            else
                fresh.cols[j].push_back((int8_t)(sb * (pb + 1)));
            da[ia] = 0;
            db[ib] = 0;
            s.st.match_sum++;
        }
        da.erase(std::remove(da.begin(), da.end(), (int8_t)0), da.end());
        if (ra != rb) db.erase(std::remove(db.begin(), db.end(), (int8_t)0), db.end());
    }
    s.expr.push_back(std::move(fresh));
}

// ---------------------------------------------------------------- state_opr.cc:285-345
static void refresh_table(State &s, const PairKey &p) {
    int64_t a = p.id0, b = p.id1;
    s.table.erase(std::remove_if(s.table.begin(), s.table.end(),
                                 [&](const Entry &e) {
                                     const PairKey &k = e.first;
                                     return k.id0 == a || k.id0 == b || k.id1 == a || k.id1 == b;
                                 }),
                  s.table.end());
    int64_t n_rows = (int64_t)s.expr.size();
    std::vector<int64_t> mod = {n_rows - 1, a};
    if (a != b) mod.push_back(b);
    std::vector<PairKey> raw;
    for (int64_t j = 0; j < s.n_out; ++j)
        for (int64_t r = 0; r < n_rows; ++r)
            for (int64_t m : mod) {
                if ((r == n_rows - 1 || r == a || r == b) && m > r) continue;
                int64_t lo = std::min(m, r), hi = std::max(m, r);
                emit_pairs(raw, lo, hi, s.expr[lo].cols[j], s.expr[hi].cols[j]);
            }
    s.st.regen_pairs += (int64_t)raw.size();
    table_batch_add(s.table, raw);
}

// optional per-iteration trace (instrumentation only; enabled by env ORC_TRACE=<file>)
static FILE *trace_fp = nullptr;
static std::chrono::steady_clock::time_point trace_t0;
static void trace_iteration(const State &s, const PairKey &p, int64_t matches) {
    int64_t it = s.st.iterations;
    int64_t live = -1, digits = -1;
    if (it % 64 == 1) {
        live = digits = 0;
        for (auto &e : s.expr) {
            int64_t d = 0;
            for (auto &c : e.cols) d += (int64_t)c.size();
            live += d > 0;
            digits += d;
        }
    }
    double now = std::chrono::duration<double>(std::chrono::steady_clock::now() - trace_t0).count();
    std::fprintf(trace_fp, "%lld %zu %lld %lld %lld %lld %lld %d %d %.3f\n", (long long)it, s.table.size(), (long long)matches,
                 (long long)live, (long long)digits, (long long)p.id0, (long long)p.id1, (int)p.shift, (int)p.sub, now);
}

// ---------------------------------------------------------------- cmvm_core.cc:10-72
static State greedy(const float *kernel, int64_t n_in, int64_t n_out, const std::string &method,
                    const std::vector<QInt> &qints_in, const std::vector<float> &lats_in, int adder_size,
                    int carry_size) {
    std::vector<QInt> qints = qints_in;
    if (qints.empty()) qints.assign(n_in, QInt{-128.0f, 127.0f, 1.0f});
    std::vector<float> lats = lats_in;
    if (lats.empty()) lats.assign(n_in, 0.0f);
    trace_t0 = std::chrono::steady_clock::now();
    State s = create_state(kernel, n_in, n_out, qints, lats, false);
    if (const char *tp = std::getenv("ORC_TRACE")) {
        if (trace_fp) std::fclose(trace_fp);
        trace_fp = std::fopen(tp, "w");
    }
    while (true) {
        if (s.table.empty()) break;
        PairKey pick;
        if (method == "mc")
            pick = pick_mc(s);
        else if (method == "mc-dc")
            pick = pick_mc_dc(s, true);
        else if (method == "mc-pdc")
            pick = pick_mc_dc(s, false);
        else if (method == "wmc")
            pick = pick_wmc(s);
        else if (method == "wmc-dc")
            pick = pick_wmc_dc(s, true);
        else if (method == "wmc-pdc")
            pick = pick_wmc_dc(s, false);
        else if (method == "dummy")
            break;
        else
            throw std::runtime_error("Unknown method: " + method);
        if (pick.id0 == -1 || pick.id1 == -1) break;
        s.st.iterations++;
        s.st.f_sum += (int64_t)s.table.size();
        s.st.f_max = std::max(s.st.f_max, (int64_t)s.table.size());
        int64_t m_before = s.st.match_sum;
        substitute(s, pick, adder_size, carry_size);
        refresh_table(s, pick);
        if (trace_fp) trace_iteration(s, pick, s.st.match_sum - m_before);
    }
    if (trace_fp) std::fflush(trace_fp);
    return s;
}

// ---------------------------------------------------------------- cmvm_core.cc:75-225
struct HeapItem {
    float lat;
    int64_t sub, align;
    float qlo, qhi, qstep;
    int64_t id, shift;
    auto key() const { return std::tie(lat, sub, align, qlo, qhi, qstep, id, shift); }
    bool operator>(const HeapItem &o) const { return key() > o.key(); }
};
static inline int64_t int_log2_trunc(const QInt &q) {
    return (int64_t)std::log2(std::max(std::abs(q.hi + q.step), std::abs(q.lo)));
}
static Stage finalize(const State &s, int adder_size, int carry_size) {
    Stage out;
    out.n_in = s.n_in;
    out.n_out = s.n_out;
    out.ops = s.ops;
    out.inp_shifts.assign(s.shift0.begin(), s.shift0.end());
    out.carry_size = carry_size;
    out.adder_size = adder_size;
    out.st = s.st;
    int64_t next_id = (int64_t)out.ops.size();
    for (int64_t j = 0; j < s.n_out; ++j) {
        std::vector<int64_t> ids, shifts, negs;
        for (size_t r = 0; r < s.expr.size(); ++r)
            for (int8_t v : s.expr[r].cols[j]) {
                ids.push_back((int64_t)r);
                shifts.push_back(dpos(v));
                negs.push_back(dsgn(v) == -1 ? 1 : 0);
            }
        if (ids.size() == 1) {
            out.out_shifts.push_back((int64_t)s.shift1[j] + shifts[0]);
            out.out_idxs.push_back(ids[0]);
            out.out_negs.push_back(negs[0]);
            continue;
        }
        if (ids.empty()) {
            out.out_idxs.push_back(-1);
            out.out_shifts.push_back((int64_t)s.shift1[j]);
            out.out_negs.push_back(0);
            continue;
        }
        std::priority_queue<HeapItem, std::vector<HeapItem>, std::greater<HeapItem>> heap;
        for (size_t k = 0; k < ids.size(); ++k) {
            const OpRec &o = out.ops[ids[k]];
            heap.push(HeapItem{o.latency, negs[k], int_log2_trunc(o.q) + shifts[k], o.q.lo, o.q.hi, o.q.step, ids[k],
                               shifts[k]});
        }
        while (heap.size() > 1) {
            HeapItem e0 = heap.top();
            heap.pop();
            HeapItem e1 = heap.top();
            heap.pop();
            QInt q0{e0.qlo, e0.qhi, e0.qstep}, q1{e1.qlo, e1.qhi, e1.qstep};
            OpRec op;
            int64_t keep_shift;
            if (e0.sub) {  // first popped term is negative: anchor on the second (cmvm_core.cc:164-174)
                int64_t sh = e0.shift - e1.shift;
                QInt q = qint_add(q1, q0, sh, e1.sub != 0, e0.sub != 0);
                auto [dl, dc] = cost_add(q1, q0, sh, (1 ^ e1.sub) != 0, adder_size, carry_size);
                op = OpRec{e1.id, e0.id, 1 ^ e1.sub, sh, q, std::max(e0.lat, e1.lat) + dl, dc};
                keep_shift = e1.shift;
            } else {
                int64_t sh = e1.shift - e0.shift;
                QInt q = qint_add(q0, q1, sh, e0.sub != 0, e1.sub != 0);
                auto [dl, dc] = cost_add(q0, q1, sh, e1.sub != 0, adder_size, carry_size);
                op = OpRec{e0.id, e1.id, e1.sub, sh, q, std::max(e0.lat, e1.lat) + dl, dc};
                keep_shift = e0.shift;
            }
            heap.push(HeapItem{op.latency, e0.sub & e1.sub, int_log2_trunc(op.q) + keep_shift, op.q.lo, op.q.hi,
                               op.q.step, next_id, keep_shift});
            out.ops.push_back(op);
            out.st.tree_ops++;
            next_id++;
        }
        HeapItem last = heap.top();
        out.out_idxs.push_back(next_id - 1);
        out.out_negs.push_back(last.sub);
        out.out_shifts.push_back((int64_t)s.shift1[j] + last.shift);
    }
    return out;
}

// cmvm_core.cc:227-237
static Stage solve_single(const float *kernel, int64_t n_in, int64_t n_out, const std::string &method,
                          const std::vector<QInt> &qints, const std::vector<float> &lats, int adder_size,
                          int carry_size) {
    State s = greedy(kernel, n_in, n_out, method, qints, lats, adder_size, carry_size);
    return finalize(s, adder_size, carry_size);
}

// ---------------------------------------------------------------- mat_decompose.cc:6-60
static std::vector<std::pair<int32_t, int32_t>> prim_mst(const std::vector<int64_t> &cost, size_t N, int dc) {
    auto lat_of = [&](size_t i, size_t j) -> float {
        return std::ceil(std::log2((float)std::max<int64_t>(cost[i * N + j], 1)));
    };
    std::vector<int32_t> parent(N, -2), latency(N, 0);
    parent[0] = -1;
    std::vector<std::pair<int32_t, int32_t>> edges;
    float cap = -1;
    if (dc >= 0) {
        int64_t m = cost[0];
        for (size_t j = 1; j < N; ++j) m = std::max(m, cost[j]);
        float max_cost0 = (float)m;
        cap = (float)((std::pow(2.0, dc) - 1) + std::ceil(std::log2(max_cost0 + 1e-32)));
    }
    for (size_t step = 1; step < N; ++step) {
        std::vector<size_t> todo, done;
        for (size_t i = 0; i < N; ++i) (parent[i] != -2 ? done : todo).push_back(i);
        int64_t best = std::numeric_limits<int64_t>::max();
        size_t bi = 0, bj = 0;
        for (size_t ii = 0; ii < todo.size(); ++ii)
            for (size_t jj = 0; jj < done.size(); ++jj) {
                size_t i = todo[ii], j = done[jj];
                int64_t c = cost[i * N + j];
                if (dc >= 0) {
                    float reach = std::max(lat_of(i, j), (float)latency[j]) + 1;
                    if (reach > cap) c = std::numeric_limits<int64_t>::max() / 2;
                }
                if (c < best) {
                    best = c;
                    bi = ii;
                    bj = jj;
                }
            }
        size_t i = todo[bi], j = done[bj];
        parent[i] = (int32_t)j;
        edges.emplace_back((int32_t)j, (int32_t)i);
        latency[i] = (int32_t)(std::max(lat_of(i, j), (float)latency[j]) + 1);
    }
    return edges;
}

// ---------------------------------------------------------------- mat_decompose.cc:63-137
static int nnz_csd(int32_t x, int N) {
    int8_t d[40];
    csd_digits(x, N, d);
    int c = 0;
    for (int n = 0; n < N; ++n) c += d[n] != 0;
    return c;
}
static void kernel_decompose(const float *kernel, int64_t n_in, int64_t n_out, int dc, std::vector<float> &m0,
                             std::vector<float> &m1) {
    std::vector<float> c(kernel, kernel + n_in * n_out);
    std::vector<int8_t> s0, s1;
    center(c, n_in, n_out, s0, s1);
    size_t W = (size_t)n_out + 1;
    std::vector<float> aug((size_t)n_in * W, 0.0f);
    for (int64_t i = 0; i < n_in; ++i)
        for (int64_t j = 0; j < n_out; ++j) aug[i * W + j + 1] = c[i * n_out + j];
    // global digit widths of the difference / sum tensors (bit_decompose.cc:23-27)
    int32_t mx0 = 0, mx1 = 0;
    for (int64_t i = 0; i < n_in; ++i)
        for (size_t a = 0; a < W; ++a)
            for (size_t b = 0; b < W; ++b) {
                mx0 = std::max(mx0, (int32_t)std::abs((int32_t)(aug[i * W + a] - aug[i * W + b])));
                mx1 = std::max(mx1, (int32_t)std::abs((int32_t)(aug[i * W + a] + aug[i * W + b])));
            }
    int N0 = csd_width(mx0), N1 = csd_width(mx1);
    std::vector<int64_t> d0(W * W, 0), d1(W * W, 0);
    for (int64_t i = 0; i < n_in; ++i)
        for (size_t a = 0; a < W; ++a)
            for (size_t b = 0; b < W; ++b) {
                d0[a * W + b] += nnz_csd((int32_t)(aug[i * W + a] - aug[i * W + b]), N0);
                d1[a * W + b] += nnz_csd((int32_t)(aug[i * W + a] + aug[i * W + b]), N1);
            }
    std::vector<int64_t> sign(W * W), dist(W * W);
    for (size_t k = 0; k < W * W; ++k) {
        sign[k] = (d1[k] - d0[k] < 0) ? -1 : 1;
        dist[k] = std::min(d0[k], d1[k]);
    }
    auto edges = prim_mst(dist, W, dc);
    m0.assign((size_t)n_in * n_out, 0.0f);
    m1.assign((size_t)n_out * n_out, 0.0f);
    if (dc == -1) {
        for (int64_t i = 0; i < n_in; ++i)
            for (int64_t j = 0; j < n_out; ++j) m0[i * n_out + j] = c[i * n_out + j];
        for (int64_t j = 0; j < n_out; ++j) m1[j * n_out + j] = 1.0f;
    } else {
        size_t cnt = 0;
        std::vector<float> col0(n_in), col1(n_out);
        for (auto [from, to] : edges) {
            float sg = (float)sign[(size_t)to * W + from];
            bool any = false;
            for (int64_t i = 0; i < n_in; ++i) {
                col0[i] = aug[i * W + to] - aug[i * W + from] * sg;
                any |= col0[i] != 0.0f;
            }
            if (from != 0)
                for (int64_t r = 0; r < n_out; ++r) col1[r] = m1[r * n_out + (from - 1)] * sg;
            else
                std::fill(col1.begin(), col1.end(), 0.0f);
            if (any) {
                col1[cnt] = 1.0f;
                for (int64_t i = 0; i < n_in; ++i) m0[i * n_out + cnt] = col0[i];
                cnt++;
            }
            for (int64_t r = 0; r < n_out; ++r) m1[r * n_out + (to - 1)] = col1[r];
        }
    }
    for (int64_t i = 0; i < n_in; ++i) {
        float sc = std::pow(2.0f, (float)s0[i]);
        for (int64_t j = 0; j < n_out; ++j) m0[i * n_out + j] *= sc;
    }
    for (int64_t j = 0; j < n_out; ++j) {
        float sc = std::pow(2.0f, (float)s1[j]);
        for (int64_t r = 0; r < n_out; ++r) m1[r * n_out + j] *= sc;
    }
}

// ---------------------------------------------------------------- api.cc:11-26
static float minimal_latency(const float *kernel, int64_t n_in, int64_t n_out, const std::vector<QInt> &qints,
                             const std::vector<float> &lats, int carry_size, int adder_size) {
    State s = create_state(kernel, n_in, n_out, qints, lats, true);
    Stage sol = finalize(s, adder_size, carry_size);
    float top = 0.0f;
    for (auto idx : sol.out_idxs) top = std::max(top, idx >= 0 ? sol.ops[idx].latency : 0.0f);
    return top;
}

static bool ends_with(const std::string &s, const char *suf) {
    size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// ---------------------------------------------------------------- api.cc:28-145
static Pipe solve_one(const float *kernel, int64_t n_in, int64_t n_out, std::string method0, std::string method1,
                      int hard_dc, int decompose_dc, const std::vector<QInt> &qints_in,
                      const std::vector<float> &lats_in, int adder_size, int carry_size) {
    if (method1 == "auto") method1 = (hard_dc >= 6 || ends_with(method0, "dc")) ? method0 : method0 + "-dc";
    if (hard_dc == 0 && !ends_with(method0, "dc")) method0 += "-dc";
    std::vector<QInt> qints = qints_in;
    if (qints.empty()) qints.assign(n_in, QInt{-128.0f, 127.0f, 1.0f});
    std::vector<float> lats = lats_in;
    if (lats.empty()) lats.assign(n_in, 0.0f);

    float min_lat = std::numeric_limits<float>::infinity();
    if (hard_dc >= 0) min_lat = minimal_latency(kernel, n_in, n_out, qints, lats, carry_size, adder_size);
    float allowed = hard_dc + min_lat;
    int log2_n = (int)std::ceil(std::log2((float)n_in));
    decompose_dc = decompose_dc == -2 ? std::min(hard_dc, log2_n) : std::min({hard_dc, decompose_dc, log2_n});

    Stage sol0, sol1;
    while (true) {
        if (decompose_dc < 0 && hard_dc >= 0) {
            if (method0 != "dummy")
                method0 = method1 = "wmc-dc";
            else
                method0 = method1 = "dummy";
        }
        std::vector<float> m0, m1;
        kernel_decompose(kernel, n_in, n_out, decompose_dc, m0, m1);
        sol0 = solve_single(m0.data(), n_in, n_out, method0, qints, lats, adder_size, carry_size);
        std::vector<float> lats0;
        std::vector<QInt> qints0;
        float top0 = 0.0f;
        for (auto idx : sol0.out_idxs) {
            float l = idx >= 0 ? sol0.ops[idx].latency : 0.0f;
            lats0.push_back(l);
            top0 = std::max(top0, l);
            qints0.push_back(idx >= 0 ? sol0.ops[idx].q : QInt{0.0f, 0.0f, std::numeric_limits<float>::infinity()});
        }
        bool both_wmc_dc = method0 == "wmc-dc" && method1 == "wmc-dc";
        if (top0 > allowed && (!both_wmc_dc || decompose_dc >= 0)) {
            decompose_dc--;
            continue;
        }
        sol1 = solve_single(m1.data(), n_out, n_out, method1, qints0, lats0, adder_size, carry_size);
        float top1 = 0.0f;
        for (auto idx : sol1.out_idxs) top1 = std::max(top1, idx >= 0 ? sol1.ops[idx].latency : 0.0f);
        if (top1 > allowed && (!both_wmc_dc || decompose_dc >= 0)) {
            decompose_dc--;
            continue;
        }
        break;
    }
    Pipe p;
    p.stages.push_back(std::move(sol0));
    p.stages.push_back(std::move(sol1));
    return p;
}

// ---------------------------------------------------------------- api.cc:147-250
static Pipe solve(const float *kernel, int64_t n_in, int64_t n_out, const std::string &method0,
                  const std::string &method1, int hard_dc, int decompose_dc, const std::vector<QInt> &qints_in,
                  const std::vector<float> &lats_in, int adder_size, int carry_size, bool search_all,
                  int *picked = nullptr) {
    std::vector<QInt> qints = qints_in;
    if (qints.empty()) qints.assign(n_in, QInt{-128.0f, 127.0f, 1.0f});
    std::vector<float> lats = lats_in;
    if (lats.empty()) lats.assign(n_in, 0.0f);
    if (picked) *picked = -1;
    if (!search_all)
        return solve_one(kernel, n_in, n_out, method0, method1, hard_dc, decompose_dc, qints, lats, adder_size,
                         carry_size);
    int hdc = hard_dc < 0 ? 1000000000 : hard_dc;
    int top_dc = std::min(hdc, (int)std::ceil(std::log2((float)n_in)));
    std::vector<int> tries;
    for (int d = -1; d <= top_dc; ++d) tries.push_back(d);
    size_t n = tries.size();
    std::vector<Pipe> cand(n);
    std::vector<float> costs(n);
    std::string err;
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < n; ++i) {
        try {
            Pipe p = solve_one(kernel, n_in, n_out, method0, method1, hdc, tries[i], qints, lats, adder_size,
                               carry_size);
            float c = 0.0f;
            for (auto &st : p.stages)
                for (auto &op : st.ops) c += op.cost;
            cand[i] = std::move(p);
            costs[i] = c;
        } catch (const std::exception &e) {
#pragma omp critical
            if (err.empty()) err = e.what();
        }
    }
    if (!err.empty()) throw std::runtime_error(err);
    size_t best = 0;
    for (size_t i = 1; i < n; ++i)
        if (costs[i] < costs[best]) best = i;
    if (picked) *picked = (int)best;
    return cand[best];
}

}  // namespace orc

// ================================================================= C ABI (ctypes)
using namespace orc;

static thread_local std::string g_err;

extern "C" {

const char *orc_last_error() { return g_err.c_str(); }

int orc_get_lsb_loc(float x) { return lsb_loc(x); }
int orc_iceil_log2(float x) { return iceil_log2(x); }
void orc_cost_add(const float *q0, const float *q1, int64_t shift, int sub, int adder_size, int carry_size,
                  float *out2) {
    auto r = cost_add(QInt{q0[0], q0[1], q0[2]}, QInt{q1[0], q1[1], q1[2]}, shift, sub != 0, adder_size, carry_size);
    out2[0] = r.first;
    out2[1] = r.second;
}
void orc_qint_add(const float *q0, const float *q1, int64_t shift, int sub0, int sub1, float *out3) {
    QInt r = qint_add(QInt{q0[0], q0[1], q0[2]}, QInt{q1[0], q1[1], q1[2]}, shift, sub0 != 0, sub1 != 0);
    out3[0] = r.lo;
    out3[1] = r.hi;
    out3[2] = r.step;
}

// digits of an int32 array; returns N. Call with out == NULL to query N.
int orc_int_arr_to_csd(const int32_t *x, int64_t n, int8_t *out) {
    int32_t mx = 0;
    for (int64_t i = 0; i < n; ++i) mx = std::max(mx, (int32_t)std::abs(x[i]));
    int N = csd_width(mx);
    if (out)
        for (int64_t i = 0; i < n; ++i) csd_digits(x[i], N, out + i * N);
    return N;
}

// returns N; csd may be NULL to query N only.
int orc_csd_decompose(const float *kernel, int64_t n_in, int64_t n_out, int do_center, int8_t *csd, int8_t *s0,
                      int8_t *s1) {
    std::vector<int8_t> c, a, b;
    int N = csd_decompose(kernel, n_in, n_out, do_center != 0, c, a, b);
    if (csd) std::memcpy(csd, c.data(), c.size());
    if (s0) std::memcpy(s0, a.data(), a.size());
    if (s1) std::memcpy(s1, b.data(), b.size());
    return N;
}

void orc_kernel_decompose(const float *kernel, int64_t n_in, int64_t n_out, int dc, float *m0, float *m1) {
    std::vector<float> a, b;
    kernel_decompose(kernel, n_in, n_out, dc, a, b);
    std::memcpy(m0, a.data(), a.size() * 4);
    std::memcpy(m1, b.data(), b.size() * 4);
}

struct OrcResult {
    Pipe pipe;
    int picked;
};

void *orc_solve(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1,
                int hard_dc, int decompose_dc, const float *qints3, const float *lats, int adder_size, int carry_size,
                int search_all) {
    try {
        std::vector<QInt> q;
        std::vector<float> l;
        if (qints3)
            for (int64_t i = 0; i < n_in; ++i) q.push_back(QInt{qints3[3 * i], qints3[3 * i + 1], qints3[3 * i + 2]});
        if (lats) l.assign(lats, lats + n_in);
        auto *r = new OrcResult;
        r->pipe = solve(kernel, n_in, n_out, method0, method1, hard_dc, decompose_dc, q, l, adder_size, carry_size,
                        search_all != 0, &r->picked);
        return r;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

// one greedy chain + adder tree on a given matrix (cmvm_core.cc:227-237), returned as a 1-stage result
void *orc_solve_single(const float *kernel, int64_t n_in, int64_t n_out, const char *method, const float *qints3,
                       const float *lats, int adder_size, int carry_size) {
    try {
        std::vector<QInt> q;
        std::vector<float> l;
        if (qints3)
            for (int64_t i = 0; i < n_in; ++i) q.push_back(QInt{qints3[3 * i], qints3[3 * i + 1], qints3[3 * i + 2]});
        else
            q.assign(n_in, QInt{-128.0f, 127.0f, 1.0f});
        if (lats)
            l.assign(lats, lats + n_in);
        else
            l.assign(n_in, 0.0f);
        auto *r = new OrcResult;
        r->picked = -1;
        r->pipe.stages.push_back(solve_single(kernel, n_in, n_out, method, q, l, adder_size, carry_size));
        return r;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

// Bounded CPU timing sample for bench.py: build the state of one greedy chain, then run greedy iterations until
// `budget_s` seconds are spent.  out[0] = seconds for state + table creation, out[1] = iterations completed,
// out[2] = seconds spent in those iterations, out[3] = 1 if the chain finished within the budget.
int orc_sample_chain(const float *kernel, int64_t n_in, int64_t n_out, const char *method, double budget_s, double *out) {
    try {
        std::vector<QInt> q(n_in, QInt{-128.0f, 127.0f, 1.0f});
        std::vector<float> l(n_in, 0.0f);
        std::string m(method);
        auto t0 = std::chrono::steady_clock::now();
        State s = create_state(kernel, n_in, n_out, q, l, false);
        auto t1 = std::chrono::steady_clock::now();
        out[0] = std::chrono::duration<double>(t1 - t0).count();
        out[1] = out[2] = out[3] = 0;
        while (true) {
            if (s.table.empty()) {
                out[3] = 1;
                break;
            }
            PairKey pick = m == "mc" ? pick_mc(s) : m == "wmc" ? pick_wmc(s) : m == "wmc-dc" ? pick_wmc_dc(s, true) : pick_mc_dc(s, true);
            if (pick.id0 == -1 || pick.id1 == -1) {
                out[3] = 1;
                break;
            }
            substitute(s, pick, -1, -1);
            refresh_table(s, pick);
            out[1] += 1;
            out[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            if (out[2] >= budget_s) break;
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int orc_n_stages(void *h) { return (int)((OrcResult *)h)->pipe.stages.size(); }
int orc_picked(void *h) { return ((OrcResult *)h)->picked; }

// info[0..4] = n_in, n_out, n_ops, carry_size, adder_size
void orc_stage_info(void *h, int s, int64_t *info) {
    const Stage &st = ((OrcResult *)h)->pipe.stages[s];
    info[0] = st.n_in;
    info[1] = st.n_out;
    info[2] = (int64_t)st.ops.size();
    info[3] = st.carry_size;
    info[4] = st.adder_size;
}
// stats[0..9]: iterations, p_init, d0, f_first, f_sum, f_max, live_sum, match_sum, regen_pairs, tree_ops
void orc_stage_stats(void *h, int s, int64_t *stats) {
    const Stats &t = ((OrcResult *)h)->pipe.stages[s].st;
    int64_t v[10] = {t.iterations, t.p_init,   t.d0,        t.f_first,     t.f_sum,
                     t.f_max,      t.live_sum, t.match_sum, t.regen_pairs, t.tree_ops};
    std::memcpy(stats, v, sizeof v);
}
// ops_i: [n_ops,4] id0,id1,opcode,data ; ops_f: [n_ops,5] lo,hi,step,latency,cost
void orc_stage_copy(void *h, int s, int64_t *inp_shifts, int64_t *out_idxs, int64_t *out_shifts, int64_t *out_negs,
                    int64_t *ops_i, float *ops_f) {
    const Stage &st = ((OrcResult *)h)->pipe.stages[s];
    std::memcpy(inp_shifts, st.inp_shifts.data(), st.inp_shifts.size() * 8);
    std::memcpy(out_idxs, st.out_idxs.data(), st.out_idxs.size() * 8);
    std::memcpy(out_shifts, st.out_shifts.data(), st.out_shifts.size() * 8);
    std::memcpy(out_negs, st.out_negs.data(), st.out_negs.size() * 8);
    for (size_t k = 0; k < st.ops.size(); ++k) {
        const OpRec &o = st.ops[k];
        ops_i[4 * k] = o.id0;
        ops_i[4 * k + 1] = o.id1;
        ops_i[4 * k + 2] = o.opcode;
        ops_i[4 * k + 3] = o.data;
        ops_f[5 * k] = o.q.lo;
        ops_f[5 * k + 1] = o.q.hi;
        ops_f[5 * k + 2] = o.q.step;
        ops_f[5 * k + 3] = o.latency;
        ops_f[5 * k + 4] = o.cost;
    }
}
void orc_free(void *h) { delete (OrcResult *)h; }

}  // extern "C"
