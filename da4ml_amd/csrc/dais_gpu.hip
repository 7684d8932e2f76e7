// dais_gpu.hip -- device executor of DAIS programs for gfx950: the samples of CombLogic.predict are independent, so one
// thread runs one sample through the whole (sequential) program.
//
// Layout: the int64 register file is [slot][sample] in HBM, so the 64 lanes of a wavefront touch 512 consecutive bytes
// per register access (fully coalesced) and every step costs 8 B written + 8 B per operand read per sample -- the
// algorithmic traffic; the kernel is HBM/L2-bandwidth bound.  Registers are *slots*: a linear
// liveness scan on the host lets a step reuse the slot of a value whose last reader has passed, so the live working set
// per sample is the maximum number of simultaneously live values (hundreds for an adder graph of 10^4-10^5 steps), not
// n_ops -- small enough for a tile of 10^5-10^6 samples to stay L2/MALL resident between producer and consumer steps.
// The decoded steps are uniform across the grid: the compiler reads them with scalar loads, the switch is a uniform
// branch.  Per-value arithmetic is dais::eval of dais_core.h, the same code the host executor is tested with.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "dais_core.h"

namespace {

#define DAIS_HIP(expr)                                                                                                 \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            throw std::runtime_error(std::string("HIP error in DAIS executor: ") + hipGetErrorString(e_) + " (" #expr ")"); \
    } while (0)

struct DevStep {
    dais::Step s;              // operand fields a/b/c hold SLOTS here (K_INPUT: a = input number)
    int32_t dst;               // slot written
    int32_t tab_off, tab_len;  // K_LUT: the table inside the concatenated table array
    int32_t pad;
};

struct DevOut {
    int32_t slot;  // -1: absent output
    int32_t neg;
    double scale;
};

constexpr int TPB = 256;

// regs: [n_slots][tile]; x: [n][n_in]; y: [n][n_out] for the samples of this tile
__global__ __launch_bounds__(TPB) void k_dais_run(const DevStep *__restrict__ steps, int64_t n_ops, const DevOut *__restrict__ outs, int64_t n_in,
                                                  int64_t n_out, const int32_t *__restrict__ tables, const double *__restrict__ x,
                                                  double *__restrict__ y, int64_t *__restrict__ regs, int64_t tile, int64_t n,
                                                  unsigned long long *__restrict__ lut_fault) {
    const int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    int64_t *R = regs + t;
    const double *xs = x + t * n_in;
    for (int64_t i = 0; i < n_ops; ++i) {
        const DevStep &d = steps[i];  // uniform address: scalar loads
        const dais::Step &s = d.s;
        const int rd = dais::reads(s.kind);
        const int64_t a = (rd & 1) ? R[(int64_t)s.a * tile] : 0;
        const int64_t b = (rd & 2) ? R[(int64_t)s.b * tile] : 0;
        const int64_t c = (rd & 4) ? R[(int64_t)s.c * tile] : 0;
        int64_t v;
        if (s.kind == dais::K_LUT) {
            const int64_t idx = dais::lut_index(s, a);
            if (idx < 0 || idx >= d.tab_len) {
                // record the first offending index and keep going with 0; the host turns it into the reference's error
                atomicCAS(lut_fault, 0ull, (1ull << 63) | ((unsigned long long)(uint32_t)d.tab_len << 32) | (unsigned long long)(uint32_t)idx);
                v = 0;
            } else
                v = tables[d.tab_off + idx];
        } else
            v = dais::eval(s, a, b, c, s.kind == dais::K_INPUT ? xs[s.a] : 0.0);
        R[(int64_t)d.dst * tile] = v;
    }
    double *ys = y + t * n_out;
    for (int64_t j = 0; j < n_out; ++j) {
        const DevOut o = outs[j];
        if (o.slot < 0) {
            ys[j] = 0.0;
            continue;
        }
        const int64_t v = R[(int64_t)o.slot * tile];
        ys[j] = (double)(o.neg ? -v : v) * o.scale;
    }
}

template <class T> struct DevBuf {
    T *p = nullptr;
    explicit DevBuf(size_t n) { DAIS_HIP(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T))); }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

}  // namespace

// Slot assignment: value i gets a slot when it is produced and gives it back after its last reader; a step may write
// the slot of an operand that dies in it (a thread reads its operands before it writes).  Outputs stay live to the end.
// Returns the number of slots; rewrites the operand fields of `steps` from value numbers to slots, fills slot_of.
// Also used by the host-side run of the device code path (da_dais_run_on with DA_DAIS_HOST_SCALAR, dais_interp.cc), which is
// how the CPU tests check the slot logic against the golden vectors.
int64_t dais_assign_slots(const dais::Program &g, std::vector<dais::Step> &steps, std::vector<int32_t> &dst, std::vector<int32_t> &slot_of) {
    const int64_t n = g.n_ops;
    std::vector<int64_t> last(n, -1);
    for (int64_t i = 0; i < n; ++i) {
        const dais::Step &s = g.steps[i];
        const int rd = dais::reads(s.kind);
        if (rd & 1) last[s.a] = i;
        if (rd & 2) last[s.b] = i;
        if (rd & 4) last[s.c] = i;
    }
    for (int32_t o : g.out_idx)
        if (o >= 0) last[o] = n;  // read by the output stage
    slot_of.assign(n, -1);
    dst.assign(n, -1);
    std::vector<int32_t> free_slots;
    int64_t n_slots = 0;
    steps = g.steps;
    for (int64_t i = 0; i < n; ++i) {
        dais::Step &s = steps[i];
        const int rd = dais::reads(s.kind);
        const int32_t ops[3] = {(rd & 1) ? s.a : -1, (rd & 2) ? s.b : -1, (rd & 4) ? s.c : -1};
        if (rd & 1) s.a = slot_of[ops[0]];
        if (rd & 2) s.b = slot_of[ops[1]];
        if (rd & 4) s.c = slot_of[ops[2]];
        for (int k = 0; k < 3; ++k) {  // operands whose last reader is this step release their slot (once per value)
            const int32_t v = ops[k];
            if (v < 0 || last[v] != i) continue;
            bool dup = false;
            for (int m = 0; m < k; ++m) dup |= ops[m] == v;
            if (!dup) free_slots.push_back(slot_of[v]);
        }
        int32_t slot;
        if (!free_slots.empty()) {
            slot = free_slots.back();
            free_slots.pop_back();
        } else
            slot = (int32_t)n_slots++;
        slot_of[i] = dst[i] = slot;
        if (last[i] < 0) free_slots.push_back(slot);  // never read: the slot is free again right away
    }
    return std::max<int64_t>(n_slots, 1);
}

void dais_run_gpu(const dais::Program &g, const double *inputs, int64_t n_samples, double *outputs) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)
        throw std::runtime_error("no HIP device: the DAIS device executor needs a GPU (the host executor does not)");
    std::vector<dais::Step> slot_steps;
    std::vector<int32_t> dst, slot_of;
    const int64_t n_slots = dais_assign_slots(g, slot_steps, dst, slot_of);
    std::vector<DevStep> steps((size_t)g.n_ops);
    std::vector<int32_t> tables, off;
    for (const auto &t : g.tables) {
        off.push_back((int32_t)tables.size());
        tables.insert(tables.end(), t.begin(), t.end());
    }
    for (int64_t i = 0; i < g.n_ops; ++i) {
        DevStep d{};
        d.s = slot_steps[i], d.dst = dst[i];
        if (d.s.kind == dais::K_LUT) d.tab_off = off[d.s.aux], d.tab_len = (int32_t)g.tables[d.s.aux].size();
        steps[i] = d;
    }
    std::vector<DevOut> outs((size_t)g.n_out);
    for (int64_t j = 0; j < g.n_out; ++j) outs[j] = DevOut{g.out_idx[j] < 0 ? -1 : slot_of[g.out_idx[j]], g.out_neg[j], g.out_scale[j]};

    // tile: as many samples as fit a register-file budget (1/4 of free memory, at most 2^22 samples), whole blocks
    size_t free_b = 0, total_b = 0;
    DAIS_HIP(hipMemGetInfo(&free_b, &total_b));
    const int64_t per_sample = n_slots * 8 + (g.n_in + g.n_out) * 8;
    int64_t tile = std::min<int64_t>({n_samples, (int64_t)1 << 22, std::max<int64_t>((int64_t)(free_b / 4) / per_sample, TPB)});
    if (const char *e = std::getenv("DA4ML_DAIS_TILE"))  // experiment knob: samples per launch (cache residency vs occupancy)
        tile = std::max<int64_t>(1, std::min<int64_t>(tile, std::atoll(e)));
    tile = (tile + TPB - 1) / TPB * TPB;

    DevBuf<DevStep> d_steps(steps.size());
    DevBuf<DevOut> d_outs(outs.size());
    DevBuf<int32_t> d_tab(tables.size());
    DevBuf<int64_t> d_regs((size_t)(n_slots * tile));
    DevBuf<double> d_x((size_t)(tile * g.n_in)), d_y((size_t)(tile * g.n_out));
    DevBuf<unsigned long long> d_fault(1);
    hipStream_t st;
    DAIS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamDestroy(s); }
    } guard{st};
    DAIS_HIP(hipMemcpyAsync(d_steps.p, steps.data(), steps.size() * sizeof(DevStep), hipMemcpyHostToDevice, st));
    DAIS_HIP(hipMemcpyAsync(d_outs.p, outs.data(), outs.size() * sizeof(DevOut), hipMemcpyHostToDevice, st));
    if (!tables.empty()) DAIS_HIP(hipMemcpyAsync(d_tab.p, tables.data(), tables.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    DAIS_HIP(hipMemsetAsync(d_fault.p, 0, sizeof(unsigned long long), st));
    for (int64_t s0 = 0; s0 < n_samples; s0 += tile) {
        const int64_t n = std::min(tile, n_samples - s0);
        DAIS_HIP(hipMemcpyAsync(d_x.p, inputs + s0 * g.n_in, (size_t)(n * g.n_in) * sizeof(double), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_dais_run, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, st, d_steps.p, g.n_ops, d_outs.p, g.n_in, g.n_out, d_tab.p,
                           d_x.p, d_y.p, d_regs.p, tile, n, d_fault.p);
        DAIS_HIP(hipGetLastError());
        if (g.n_out > 0)
            DAIS_HIP(hipMemcpyAsync(outputs + s0 * g.n_out, d_y.p, (size_t)(n * g.n_out) * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    unsigned long long fault = 0;
    DAIS_HIP(hipMemcpyAsync(&fault, d_fault.p, sizeof fault, hipMemcpyDeviceToHost, st));
    DAIS_HIP(hipStreamSynchronize(st));
    if (fault) {
        const int64_t idx = (int64_t)(int32_t)(uint32_t)(fault & 0xFFFFFFFFull);
        throw std::runtime_error("Logic lookup index out of bounds: index=" + std::to_string(idx) +
                                 ", table_size=" + std::to_string((fault >> 32) & 0x7FFFFFFFull));
    }
}
