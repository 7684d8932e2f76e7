// dais_core.h -- data model and per-value arithmetic of the DAIS executors, shared by the host block executor
// (dais_interp.cc) and the device executor (dais_gpu.hip), so that both compute every operation with the same code.
// Semantics: reference docs/dais.md:25-105 and src/da4ml/_binary/dais/DAISInterpreter.cc:139-388, with every shift the
// reference derives per sample from the operand formats folded into the decoded step.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DAIS_HD __host__ __device__ __forceinline__
#else
#define DAIS_HD inline
#endif

namespace dais {

struct Fmt {  // fixed-point format (signed, integer bits, fractional bits)
    int32_t sgn, ints, frac;
    int width() const { return ints + frac + (sgn ? 1 : 0); }
};

enum Kind : int32_t { K_INPUT, K_ADDSUB, K_RELU, K_QUANT, K_CADD, K_CONST, K_MUX, K_MUL, K_LUT, K_BITU, K_BITB };

struct Step {
    int32_t kind;
    int32_t a, b, c;       // operand registers (c: mux condition); K_INPUT: a = input number
    int32_t neg;           // operand negation flags (bit 0: a, bit 1: b)
    int32_t sh_a, sh_b;    // left shifts applied to operand a / b
    int32_t sh_out;        // right shift applied to the result (drop of fractional bits)
    int32_t wrap_w;        // kinds that quantise: wrap the result into this many bits ...
    int64_t wrap_lo;       // ... starting at this minimum
    int64_t imm;           // constant / table offset / mask / msb threshold
    int32_t aux;           // table index, bit operation, "condition is signed"
    double scale;          // input scaling 2^(inp_shift + frac)
};

struct Program {
    int64_t n_in = 0, n_out = 0, n_ops = 0;
    std::vector<Step> steps;
    std::vector<std::vector<int32_t>> tables;
    std::vector<int32_t> out_idx, out_neg;
    std::vector<double> out_scale;
};

// decode + validate an int32 DAIS program (throws std::runtime_error with the reference's messages) -- dais_interp.cc
Program decode(const int32_t *p, int64_t n_words);

// two's-complement wrap of v into `w` bits whose smallest value is `lo` (reference DAISInterpreter.cc:139-152)
DAIS_HD int64_t wrap(int64_t v, int w, int64_t lo) {
    const uint64_t mask = w >= 64 ? ~0ull : ((1ull << w) - 1);
    return (int64_t)(((uint64_t)v - (uint64_t)lo) & mask) + lo;
}
DAIS_HD int64_t shl(int64_t v, int s) { return (int64_t)((uint64_t)v << s); }

// Value of one step for one sample.  a, b, c: current values of the operand registers (ignored where the kind has none);
// x: the raw input value for K_INPUT.  K_LUT is resolved by the caller (needs the table and a range check):
// `lut_index` gives the index into table `s.aux`.
DAIS_HD int64_t lut_index(const Step &s, int64_t a) { return a - s.imm; }
DAIS_HD int64_t eval(const Step &s, int64_t a, int64_t b, int64_t c, double x) {
    switch (s.kind) {
    case K_INPUT: return wrap((int64_t)floor(x * s.scale), s.wrap_w, s.wrap_lo);
    case K_ADDSUB: return ((s.neg & 2) ? shl(a, s.sh_a) - shl(b, s.sh_b) : shl(a, s.sh_a) + shl(b, s.sh_b)) >> s.sh_out;
    case K_RELU: {
        const int64_t v = s.neg ? -a : a;
        return v < 0 ? 0 : wrap(v >> s.sh_out, s.wrap_w, s.wrap_lo);
    }
    case K_QUANT: return wrap((s.neg ? -a : a) >> s.sh_out, s.wrap_w, s.wrap_lo);
    case K_CADD: return shl(a, s.sh_a) + s.imm;
    case K_CONST: return s.imm;
    case K_MUX: {
        const bool msb = s.aux ? c < 0 : c > s.imm;
        return wrap(msb ? shl(a, s.sh_a) : shl((s.neg & 2) ? -b : b, s.sh_b), s.wrap_w, s.wrap_lo);
    }
    case K_MUL: return (int64_t)((uint64_t)a * (uint64_t)b);
    case K_BITU: {
        const int64_t v = s.neg ? -a : a;
        const int op = s.aux & 15;
        return op == 0 ? ((s.aux & 16) ? ~v : (~v & s.imm)) : op == 1 ? (int64_t)(v != 0) : (int64_t)((v & s.imm) == s.imm);
    }
    case K_BITB: {
        const int64_t p = shl((s.neg & 1) ? -a : a, s.sh_a), q = shl((s.neg & 2) ? -b : b, s.sh_b);
        return s.aux == 0 ? (p & q) : s.aux == 1 ? (p | q) : (p ^ q);
    }
    default: return 0;  // K_LUT: caller
    }
}

// which operand registers a step reads: bit 0 a, bit 1 b, bit 2 c
DAIS_HD int reads(int32_t kind) {
    switch (kind) {
    case K_INPUT:
    case K_CONST: return 0;
    case K_ADDSUB:
    case K_MUL:
    case K_BITB: return 3;
    case K_MUX: return 7;
    default: return 1;
    }
}

}  // namespace dais
