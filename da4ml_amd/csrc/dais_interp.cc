// dais_interp.cc -- host-side executor of DAIS programs (the int32 program format of da4ml's CombLogic.to_binary,
// reference docs/dais.md:25-105).  It replaces the reference's C++ interpreter behind `da4ml._binary.dais_interp_run`
// (reference src/da4ml/_binary/dais/DAISInterpreter.cc:291-388 and bindings.cc:30-100) as the integer-exact functional
// checker of solver results.  Like the reference it runs on the host: a DAIS program is a
// strictly sequential chain of a few thousand scalar operations per sample; samples are independent and are split
// over host threads.
//
// Design (not a transcription of the reference's class): the program is decoded ONCE into a flat array of fully
// resolved steps -- every shift that the reference recomputes per sample from the operand formats is folded into the
// step at load time, invalid programs are rejected at load time -- and a sample is then a single pass over that array
// with one int64 register file.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/da4ml_hip.h"

#include "dais_core.h"

namespace dais {
namespace {

[[noreturn]] void bad(const std::string &msg) { throw std::runtime_error(msg); }

inline int64_t fmt_min(const Fmt &f) { return f.sgn ? -((int64_t)1 << (f.width() - 1)) : 0; }

void set_wrap(Step &s, const Fmt &to) {
    int w = to.width();
    if (w < 0) bad("operation format with a negative width");
    if (w > 62) bad("operation format wider than 62 bits is not supported");
    s.wrap_w = w;
    s.wrap_lo = w > 0 ? fmt_min(to) : 0;
}

}  // namespace

Program decode(const int32_t *p, int64_t n_words) {
    if (n_words < 6) bad("Binary data too small to contain valid DAIS model file");
    if (p[0] != 1) bad("DAIS version mismatch: expected version 1, got version " + std::to_string(p[0]));
    Program g;
    g.n_in = p[2], g.n_out = p[3], g.n_ops = p[4];
    const int64_t n_tables = p[5];
    if (g.n_in < 0 || g.n_out < 0 || g.n_ops < 0 || n_tables < 0) bad("negative size in DAIS header");
    const int64_t head = 6, code0 = head + g.n_in + 3 * g.n_out, tab0 = code0 + 8 * g.n_ops;
    int64_t expect = tab0;
    if (tab0 + n_tables > n_words) bad("Binary data size mismatch");
    for (int64_t t = 0; t < n_tables; ++t) expect += 1 + (int64_t)p[tab0 + t];
    if (expect != n_words)
        bad("Binary data size mismatch: expected " + std::to_string(expect * 4) + " bytes , got " + std::to_string(n_words * 4) + " bytes");
    const int32_t *inp_shift = p + head, *out_idx = inp_shift + g.n_in, *out_shift = out_idx + g.n_out, *out_neg = out_shift + g.n_out;
    int64_t off = tab0 + n_tables;
    for (int64_t t = 0; t < n_tables; ++t) {
        g.tables.emplace_back(p + off, p + off + p[tab0 + t]);
        off += p[tab0 + t];
    }
    std::vector<Fmt> fmt((size_t)g.n_ops);
    for (int64_t i = 0; i < g.n_ops; ++i) fmt[i] = Fmt{p[code0 + 8 * i + 5], p[code0 + 8 * i + 6], p[code0 + 8 * i + 7]};
    g.steps.resize((size_t)g.n_ops);
    auto reg = [&](int64_t i, int32_t id, const char *what) {  // causality (reference DAISInterpreter.cc:428-447)
        if (id < 0 || id >= i) bad("Operation " + std::to_string(i) + " has " + what + "=" + std::to_string(id) + " violating causality");
        return id;
    };
    auto lshift = [&](int64_t i, int32_t s) {
        if (s < 0 || s > 62) bad("Operation " + std::to_string(i) + " needs an unsupported shift of " + std::to_string(s) + " bits");
        return s;
    };
    for (int64_t i = 0; i < g.n_ops; ++i) {
        const int32_t *w = p + code0 + 8 * i;
        const int32_t opcode = w[0], id0 = w[1], id1 = w[2], lo = w[3], hi = w[4];
        const Fmt &to = fmt[i];
        Step s{};
        switch (opcode) {
        case -1:  // input copy: floor(x * 2^(inp_shift + frac)) wrapped into the op's own format
            if (id0 < 0 || id0 >= g.n_in) bad("Operation " + std::to_string(i) + " reads input " + std::to_string(id0) + " of " + std::to_string(g.n_in));
            s.kind = K_INPUT, s.a = id0, s.scale = std::pow(2.0, inp_shift[id0] + to.frac);
            set_wrap(s, to);
            break;
        case 0:
        case 1: {  // buf[id0] +/- buf[id1] * 2^lo, aligned on the finer of the two operand grids
            s.kind = K_ADDSUB, s.a = reg(i, id0, "id0"), s.b = reg(i, id1, "id1"), s.neg = opcode == 1 ? 2 : 0;
            const int actual = lo + fmt[id0].frac - fmt[id1].frac;
            s.sh_a = actual > 0 ? 0 : lshift(i, -actual);
            s.sh_b = actual > 0 ? lshift(i, actual) : 0;
            const int drop = std::max(fmt[id0].frac, fmt[id1].frac - lo) - to.frac;
            s.sh_out = drop > 0 ? lshift(i, drop) : 0;
            break;
        }
        case 2:
        case -2:
        case 3:
        case -3:  // relu / quantize of +/- buf[id0]: drop fractional bits, wrap
            s.kind = (opcode == 2 || opcode == -2) ? K_RELU : K_QUANT;
            s.a = reg(i, id0, "id0"), s.neg = opcode < 0 ? 1 : 0;
            s.sh_out = lshift(i, fmt[id0].frac - to.frac);
            set_wrap(s, to);
            break;
        case 4:  // buf[id0] rescaled to the op's grid + 64-bit immediate
            s.kind = K_CADD, s.a = reg(i, id0, "id0"), s.sh_a = lshift(i, to.frac - fmt[id0].frac);
            s.imm = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
            break;
        case 5:
            s.kind = K_CONST, s.imm = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
            break;
        case 6:
        case -6: {  // msb(buf[lo]) ? buf[id0] : +/- buf[id1] * 2^hi, then quantised to the op's format
            s.kind = K_MUX, s.a = reg(i, id0, "id0"), s.b = reg(i, id1, "id1"), s.c = reg(i, lo, "cond_idx"), s.neg = opcode < 0 ? 2 : 0;
            const int sh0 = to.frac - fmt[id0].frac, sh1 = to.frac - fmt[id1].frac + hi;
            if (sh0 != 0 && sh1 != 0) bad("Unsupported msb_mux shift configuration: shift0=" + std::to_string(sh0) + ", shift1=" + std::to_string(sh1));
            s.sh_a = lshift(i, sh0), s.sh_b = lshift(i, sh1);
            const Fmt &fc = fmt[lo];
            s.aux = fc.sgn;
            if (!fc.sgn) {
                // the reference's unsigned msb test is `value > max(1LL << (width-2), 0LL)` (DAISInterpreter.cc:177-181).
                // For a 1-bit condition -- what `np.where(cond.to_bool(), ..)` traces to -- the shift count is -1,
                // which the x86 `shl` of the reference build takes modulo 64: 1 << 63 is negative, the max() gives 0 and
                // the test is `value > 0`.  Reproduced here for every width < 2 (count & 63), not rejected.
                const int64_t t = (int64_t)((uint64_t)1 << ((unsigned)(fc.width() - 2) & 63u));
                s.imm = t > 0 ? t : 0;
            }
            set_wrap(s, to);
            break;
        }
        case 7:
            s.kind = K_MUL, s.a = reg(i, id0, "id0"), s.b = reg(i, id1, "id1");
            break;
        case 8: {  // table[lo][buf[id0] - min(format of id0) - hi]
            s.kind = K_LUT, s.a = reg(i, id0, "id0"), s.aux = lo;
            if (lo < 0 || lo >= (int32_t)g.tables.size()) bad("Operation " + std::to_string(i) + " uses lookup table " + std::to_string(lo));
            s.imm = fmt_min(fmt[id0]) + hi;
            break;
        }
        case 9:
        case -9:  // bitwise unary on +/- buf[id0]: lo = 0 NOT, 1 reduce-OR, 2 reduce-AND
            s.kind = K_BITU, s.a = reg(i, id0, "id0"), s.neg = opcode < 0 ? 1 : 0, s.aux = lo | (to.sgn ? 16 : 0);
            if (lo < 0 || lo > 2) bad("Unknown bit unary operation with data_low=" + std::to_string(lo));
            if (fmt[id0].width() < 0 || fmt[id0].width() > 62) bad("bit operation on a format wider than 62 bits");
            s.imm = ((int64_t)1 << fmt[id0].width()) - 1;
            break;
        case 10: {  // bitwise binary: (hi bit 0: -buf[id0]) op (hi bit 1: -buf[id1]) * 2^lo aligned; hi >> 24 = 0 AND, 1 OR, 2 XOR
            s.kind = K_BITB, s.a = reg(i, id0, "id0"), s.b = reg(i, id1, "id1"), s.neg = hi & 3, s.aux = hi >> 24;
            if (s.aux < 0 || s.aux > 2) bad("Unknown bit binary operation with data_low=" + std::to_string(lo));
            const int actual = lo + fmt[id0].frac - fmt[id1].frac;
            s.sh_a = actual > 0 ? 0 : lshift(i, -actual);
            s.sh_b = actual > 0 ? lshift(i, actual) : 0;
            break;
        }
        default: bad("Unknown opcode: " + std::to_string(opcode) + " at index " + std::to_string(i));
        }
        g.steps[i] = s;
    }
    g.out_idx.assign(out_idx, out_idx + g.n_out);
    g.out_neg.assign(out_neg, out_neg + g.n_out);
    g.out_scale.assign((size_t)g.n_out, 0.0);
    for (int64_t j = 0; j < g.n_out; ++j) {
        if (out_idx[j] >= g.n_ops) bad("output " + std::to_string(j) + " reads operation " + std::to_string(out_idx[j]));
        if (out_idx[j] >= 0) g.out_scale[j] = std::pow(2.0, out_shift[j] - fmt[out_idx[j]].frac);
    }
    return g;
}

namespace {

// A block of nb <= BLOCK samples: x[nb][n_in] -> y[nb][n_out].  Steps outermost, samples innermost, so that every step is
// decoded once per block and its arithmetic is a short unit-stride loop over the block (vectorised by the compiler);
// reg[n_ops][BLOCK] is the register file of the block.
constexpr int BLOCK = 8;
// host-only function multiversioning (the file also passes through hipcc's device pass, where it is not available)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define DA_CPU_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define DA_CPU_CLONES
#endif

DA_CPU_CLONES
void run_block(const Program &g, const double *x, double *y, int nb, int64_t *reg) {
    const Step *st = g.steps.data();
    for (int64_t i = 0; i < g.n_ops; ++i) {
        const Step &s = st[i];
        int64_t *__restrict r = reg + i * BLOCK;
        const int64_t *ra = reg + (int64_t)s.a * BLOCK, *rb = reg + (int64_t)s.b * BLOCK;
        switch (s.kind) {
        case K_INPUT:
            for (int k = 0; k < nb; ++k) r[k] = wrap((int64_t)std::floor(x[k * g.n_in + s.a] * s.scale), s.wrap_w, s.wrap_lo);
            break;
        case K_ADDSUB:
            if (s.neg & 2)
                for (int k = 0; k < BLOCK; ++k) r[k] = ((int64_t)((uint64_t)ra[k] << s.sh_a) - (int64_t)((uint64_t)rb[k] << s.sh_b)) >> s.sh_out;
            else
                for (int k = 0; k < BLOCK; ++k) r[k] = ((int64_t)((uint64_t)ra[k] << s.sh_a) + (int64_t)((uint64_t)rb[k] << s.sh_b)) >> s.sh_out;
            break;
        case K_RELU:
            for (int k = 0; k < BLOCK; ++k) {
                const int64_t a = s.neg ? -ra[k] : ra[k];
                r[k] = a < 0 ? 0 : wrap(a >> s.sh_out, s.wrap_w, s.wrap_lo);
            }
            break;
        case K_QUANT:
            for (int k = 0; k < BLOCK; ++k) r[k] = wrap((s.neg ? -ra[k] : ra[k]) >> s.sh_out, s.wrap_w, s.wrap_lo);
            break;
        case K_CADD:
            for (int k = 0; k < BLOCK; ++k) r[k] = (int64_t)((uint64_t)ra[k] << s.sh_a) + s.imm;
            break;
        case K_CONST:
            for (int k = 0; k < BLOCK; ++k) r[k] = s.imm;
            break;
        case K_MUX: {
            const int64_t *rc = reg + (int64_t)s.c * BLOCK;
            for (int k = 0; k < BLOCK; ++k) {
                const bool msb = s.aux ? rc[k] < 0 : rc[k] > s.imm;
                const int64_t b = (s.neg & 2) ? -rb[k] : rb[k];
                r[k] = wrap(msb ? (int64_t)((uint64_t)ra[k] << s.sh_a) : (int64_t)((uint64_t)b << s.sh_b), s.wrap_w, s.wrap_lo);
            }
            break;
        }
        case K_MUL:
            for (int k = 0; k < BLOCK; ++k) r[k] = (int64_t)((uint64_t)ra[k] * (uint64_t)rb[k]);
            break;
        case K_LUT: {
            const std::vector<int32_t> &t = g.tables[s.aux];
            for (int k = 0; k < nb; ++k) {  // only the live samples: padding lanes may hold anything
                const int64_t idx = ra[k] - s.imm;
                if (idx < 0 || idx >= (int64_t)t.size())
                    bad("Logic lookup index out of bounds: index=" + std::to_string(idx) + ", table_size=" + std::to_string(t.size()));
                r[k] = t[(size_t)idx];
            }
            break;
        }
        case K_BITU:
            for (int k = 0; k < BLOCK; ++k) {
                const int64_t a = s.neg ? -ra[k] : ra[k];
                const int op = s.aux & 15;
                r[k] = op == 0 ? ((s.aux & 16) ? ~a : (~a & s.imm)) : op == 1 ? (int64_t)(a != 0) : (int64_t)((a & s.imm) == s.imm);
            }
            break;
        default:  // K_BITB
            for (int k = 0; k < BLOCK; ++k) {
                const int64_t a = (int64_t)((uint64_t)((s.neg & 1) ? -ra[k] : ra[k]) << s.sh_a);
                const int64_t b = (int64_t)((uint64_t)((s.neg & 2) ? -rb[k] : rb[k]) << s.sh_b);
                r[k] = s.aux == 0 ? (a & b) : s.aux == 1 ? (a | b) : (a ^ b);
            }
            break;
        }
    }
    for (int64_t j = 0; j < g.n_out; ++j) {
        const int32_t o = g.out_idx[j];
        for (int k = 0; k < nb; ++k) {
            if (o < 0) {
                y[k * g.n_out + j] = 0.0;
                continue;
            }
            const int64_t v = reg[(int64_t)o * BLOCK + k];
            y[k * g.n_out + j] = (double)(g.out_neg[j] ? -v : v) * g.out_scale[j];
        }
    }
}

// One sample at a time through dais::eval over the slot-compacted steps -- the exact per-thread code path of the device
// executor (dais_gpu.hip: same eval, same slot assignment).  Slower than run_block; exists so that the shared arithmetic and
// the slot logic are pinned against the golden vectors on hosts without a GPU.
struct SlotProgram {
    std::vector<Step> steps;
    std::vector<int32_t> dst, slot_of;
    int64_t n_slots = 0;
};
void run_scalar(const Program &g, const SlotProgram &sp, const double *x, double *y, int64_t *reg) {
    for (int64_t i = 0; i < g.n_ops; ++i) {
        const Step &s = sp.steps[i];
        const int rd = reads(s.kind);
        const int64_t a = (rd & 1) ? reg[s.a] : 0, b = (rd & 2) ? reg[s.b] : 0, c = (rd & 4) ? reg[s.c] : 0;
        if (s.kind == K_LUT) {
            const std::vector<int32_t> &t = g.tables[s.aux];
            const int64_t idx = lut_index(s, a);
            if (idx < 0 || idx >= (int64_t)t.size())
                bad("Logic lookup index out of bounds: index=" + std::to_string(idx) + ", table_size=" + std::to_string(t.size()));
            reg[sp.dst[i]] = t[(size_t)idx];
        } else
            reg[sp.dst[i]] = eval(s, a, b, c, s.kind == K_INPUT ? x[s.a] : 0.0);
    }
    for (int64_t j = 0; j < g.n_out; ++j) {
        const int32_t o = g.out_idx[j];
        const int64_t v = o < 0 ? 0 : reg[sp.slot_of[o]];
        y[j] = o < 0 ? 0.0 : (double)(g.out_neg[j] ? -v : v) * g.out_scale[j];
    }
}

thread_local std::string g_dais_err;

}  // namespace
}  // namespace dais

using namespace dais;

// device executor (dais_gpu.hip)
int64_t dais_assign_slots(const dais::Program &g, std::vector<dais::Step> &steps, std::vector<int32_t> &dst, std::vector<int32_t> &slot_of);
void dais_run_gpu(const dais::Program &g, const double *inputs, int64_t n_samples, double *outputs);

extern "C" {

const char *da_dais_last_error(void) { return g_dais_err.c_str(); }

int da_dais_run(const int32_t *program, int64_t n_words, const double *inputs, int64_t n_samples, double *outputs, int n_threads) {
    return da_dais_run_on(program, n_words, inputs, n_samples, outputs, n_threads, DA_DAIS_HOST);
}

int da_dais_run_on(const int32_t *program, int64_t n_words, const double *inputs, int64_t n_samples, double *outputs, int n_threads, int where) {
    try {
        if (!program || n_words < 4) bad("Invalid binary logic data");
        const Program g = decode(program, n_words);
        if (n_samples <= 0) return DA_OK;
        if (where == DA_DAIS_DEVICE) {
            dais_run_gpu(g, inputs, n_samples, outputs);
            return DA_OK;
        }
        SlotProgram sp;
        if (where == DA_DAIS_HOST_SCALAR) sp.n_slots = dais_assign_slots(g, sp.steps, sp.dst, sp.slot_of);
        if (where != DA_DAIS_HOST && where != DA_DAIS_HOST_SCALAR) bad("da_dais_run_on: unknown executor " + std::to_string(where));
        // split like the reference (bindings.cc:57-66): at least 32 samples per thread
        int64_t hw = (int64_t)std::max(1u, std::thread::hardware_concurrency());
        int64_t want = n_threads <= 0 ? hw : std::min<int64_t>(n_threads, hw);
        int64_t per = std::max<int64_t>(n_samples / std::max<int64_t>(want, 1), 32);
        int64_t n_thr = (n_samples + per - 1) / per;
        std::exception_ptr err;
        std::mutex mu;
        auto work = [&](int64_t lo, int64_t hi) {
            try {
                std::vector<int64_t> reg((size_t)std::max<int64_t>(g.n_ops, 1) * BLOCK, 0);
                if (where == DA_DAIS_HOST_SCALAR) {
                    std::vector<int64_t> slots((size_t)sp.n_slots, 0);
                    for (int64_t s = lo; s < hi; ++s) run_scalar(g, sp, inputs + s * g.n_in, outputs + s * g.n_out, slots.data());
                    return;
                }
                for (int64_t s = lo; s < hi; s += BLOCK)
                    run_block(g, inputs + s * g.n_in, outputs + s * g.n_out, (int)std::min<int64_t>(BLOCK, hi - s), reg.data());
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu);
                if (!err) err = std::current_exception();
            }
        };
        if (n_thr <= 1)
            work(0, n_samples);
        else {
            std::vector<std::thread> th;
            for (int64_t t = 0; t < n_thr; ++t) th.emplace_back(work, t * per, std::min(n_samples, (t + 1) * per));
            for (auto &t : th) t.join();
        }
        if (err) std::rethrow_exception(err);
        return DA_OK;
    } catch (const std::exception &e) {
        g_dais_err = e.what();
        return DA_ERR_RUNTIME;
    }
}

}  // extern "C"
