// cmvm_capi.cc -- the C ABI of libda4ml_hip.so (declared in include/da4ml_hip.h).
// Every compute entry point goes through the HIP backend; there is no CPU fallback: without a usable
// device the calls fail with DA_ERR_NO_DEVICE / NULL and a message.

#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/da4ml_hip.h"
#include "cmvm_gpu.h"
#include "cmvm_host.h"
#include "cmvm_rccl.h"
#include "cmvm_shard.h"

namespace {

thread_local std::string g_err;
thread_local int g_err_code = DA_OK;  // DA_ERR_* of the calling thread's last failure (da_last_error_code)
// The library lock, held for a whole solve.  fork(): the CHILD gets a fresh, unlocked mutex (the thread that may have held the
// parent's does not exist there; its first call would hang on an inherited locked one).  Nothing is taken in the parent: a
// prepare handler that waited for the lock would stall fork() for a whole solve -- minutes -- and, called from a Python thread,
// would do so holding the GIL, which the all-reduce callback of a sharded solve running in another thread needs: a deadlock.
// A fork while ANOTHER thread is inside a solve leaves the child with library state caught mid-update: unsupported (as is any use
// of a HIP context in a forked child); forking between solves, or from inside the all-reduce callback, is fine.
struct LibraryMutex {
    std::mutex m;
    static thread_local bool mine;
    static std::atomic<bool> busy;      // some thread is inside a library call
    static std::atomic<bool> poisoned;  // this process is the child of a fork taken while ANOTHER thread was inside a call: the copied
                                        // state (backend, host pool, device buffers) was caught mid-update -- every later call says so
    void lock() {
        m.lock();
        mine = true;
        busy.store(true, std::memory_order_relaxed);
    }
    void unlock() {
        busy.store(false, std::memory_order_relaxed);
        mine = false;
        m.unlock();
    }
    LibraryMutex() {
        pthread_atfork(nullptr, nullptr, [] {
            if (!mine) {
                if (busy.load(std::memory_order_relaxed)) poisoned.store(true, std::memory_order_relaxed);
                new (&instance->m) std::mutex();  // (the inherited one may be locked by a thread that was not copied; it is never
                busy.store(false, std::memory_order_relaxed);  // destroyed.  Held by the forking thread itself: it exists in the child and goes on holding it)
            }
        });
        instance = this;
    }
    static LibraryMutex *instance;
};
thread_local bool LibraryMutex::mine = false;
std::atomic<bool> LibraryMutex::busy{false};
std::atomic<bool> LibraryMutex::poisoned{false};
LibraryMutex *LibraryMutex::instance = nullptr;
LibraryMutex g_mutex;
int g_device = 0;
std::unique_ptr<da::gpu::HipBackend> g_backend;
int g_backend_device = -1;

da::gpu::HipBackend &backend() {
    if (LibraryMutex::poisoned.load(std::memory_order_relaxed))
        throw std::runtime_error("libda4ml_hip: this process was forked while another thread was inside a solve; the copied library state is unusable here (fork between solves, or exec)");
    if (!g_backend || g_backend_device != g_device) {
        if (da::gpu::device_count() <= g_device) throw std::runtime_error("no HIP device " + std::to_string(g_device) + " available (libda4ml_hip has no CPU path)");
        g_backend.reset(new da::gpu::HipBackend(g_device));
        g_backend_device = g_device;
    }
    return *g_backend;
}

int fail(const std::exception &e) {
    g_err = e.what();
    if (dynamic_cast<const std::invalid_argument *>(&e))
        g_err_code = DA_ERR_VALUE;
    else if (g_err.rfind("no HIP device", 0) == 0)
        g_err_code = DA_ERR_NO_DEVICE;
    else
        g_err_code = DA_ERR_RUNTIME;
    return g_err_code;
}

// Steps the latency model has no logarithm for are refused up front: non-positive, NaN / infinite and subnormal ones (the
// reference computes -log2 of whatever it is given; a result with infinite or NaN latencies is of no use to anybody).  Steps
// that are not powers of two are accepted, any number of them (StepLog2, cmvm_core.h: one table row per distinct mantissa).
void check_dyadic_steps(const float *q, int64_t n_in) {
    if (!q) return;
    for (int64_t i = 0; i < n_in; ++i) {
        float lo = q[3 * i], hi = q[3 * i + 1], st = q[3 * i + 2];
        if (lo == 0.0f && hi == 0.0f) continue;  // constant-zero input: its digits are dropped, the step is never used
        uint32_t b;
        std::memcpy(&b, &st, 4);
        const uint32_t e = (b >> 23) & 0xFF;
        if ((b >> 31) || e == 0 || e == 255) throw std::invalid_argument("qintervals[" + std::to_string(i) + "].step must be a positive normal number");
    }
}

}  // namespace

struct da_result {
    da::PipeResult pipe;
    da::ChainStats stats;
};

extern "C" {

const char *da_last_error(void) { return g_err.c_str(); }
int da_last_error_code(void) { return g_err_code; }
const char *da_version(void) { return "da4ml_hip 0.1 (gfx950)"; }
int da_device_count(void) { return da::gpu::device_count(); }
int da_set_device(int device) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    if (device < 0 || device >= da::gpu::device_count()) {
        g_err = "invalid device index " + std::to_string(device);
        return g_err_code = DA_ERR_NO_DEVICE;
    }
    g_device = device;
    return DA_OK;
}

int da_get_lsb_loc(float x) { return da::lsb_loc(x); }
int da_iceil_log2(float x) { return da::iceil_log2(x); }
int da_cost_add(const float *q0, const float *q1, int64_t shift, int sub, int adder_size, int carry_size, float *out2) {
    da::cost_add(da::QInt{q0[0], q0[1], q0[2]}, da::QInt{q1[0], q1[1], q1[2]}, shift, sub != 0, adder_size, carry_size, out2[0], out2[1]);
    return DA_OK;
}

int da_int_arr_to_csd(const int32_t *x, int64_t n, int8_t *out) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    try {
        std::vector<int8_t> csd;
        int N = backend().int_to_csd(x, n, csd);
        if (out) std::memcpy(out, csd.data(), csd.size());
        return N;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int da_csd_decompose(const float *kernel, int64_t n_in, int64_t n_out, int center, int8_t *csd, int8_t *shift0, int8_t *shift1) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    try {
        std::vector<int8_t> c, a, b;
        int N = backend().csd_decompose(kernel, (int)n_in, (int)n_out, center != 0, c, a, b);
        if (csd) std::memcpy(csd, c.data(), c.size());
        if (shift0) std::memcpy(shift0, a.data(), a.size());
        if (shift1) std::memcpy(shift1, b.data(), b.size());
        return N;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int da_kernel_decompose(const float *kernel, int64_t n_in, int64_t n_out, int dc, float *m0, float *m1) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    try {
        std::vector<float> a, b;
        da::kernel_decompose(backend(), kernel, (int)n_in, (int)n_out, dc, a, b);
        std::memcpy(m0, a.data(), a.size() * 4);
        std::memcpy(m1, b.data(), b.size() * 4);
        return DA_OK;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int da_solve_batch(int count, const float *const *kernels, const int64_t *n_in, const int64_t *n_out, const char *method0,
                   const char *method1, int hard_dc, int decompose_dc, const float *const *qintervals,
                   const float *const *latencies, int adder_size, int carry_size, int search_all_decompose_dc,
                   da_result **results) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    try {
        std::vector<da::Problem> probs((size_t)count);
        for (int i = 0; i < count; ++i) {
            da::Problem &p = probs[i];
            p.kernel = kernels[i];
            p.n_in = (int)n_in[i];
            p.n_out = (int)n_out[i];
            if (p.n_in <= 0 || p.n_out <= 0) throw std::invalid_argument("kernel must be a non-empty 2-D matrix");
            p.opt.method0 = method0;
            p.opt.method1 = method1;
            p.opt.hard_dc = hard_dc;
            p.opt.decompose_dc = decompose_dc;
            const float *q = qintervals ? qintervals[i] : nullptr;
            const float *l = latencies ? latencies[i] : nullptr;
            check_dyadic_steps(q, p.n_in);
            if (q)
                for (int r = 0; r < p.n_in; ++r) p.opt.qints.push_back(da::QInt{q[3 * r], q[3 * r + 1], q[3 * r + 2]});
            if (l) p.opt.lats.assign(l, l + p.n_in);
            p.opt.adder_size = adder_size;
            p.opt.carry_size = carry_size;
            p.opt.search_all = search_all_decompose_dc != 0;
        }
        std::vector<da::ChainStats> stats;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<da::PipeResult> res = da::solve_batch(backend(), probs, &stats);
        for (int i = 0; i < count; ++i) results[i] = new da_result{std::move(res[i]), i < (int)stats.size() ? stats[i] : da::ChainStats{}};
        if (std::getenv("DA4ML_HIP_VERBOSE"))
            std::fprintf(stderr, "[da4ml_hip] da_solve_batch(%d): %.2f ms\n", count, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        return DA_OK;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

da_result *da_solve(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                    int decompose_dc, const float *qintervals, const float *latencies, int adder_size, int carry_size,
                    int search_all_decompose_dc) {
    da_result *r = nullptr;
    const float *q[1] = {qintervals}, *l[1] = {latencies};
    int rc = da_solve_batch(1, &kernel, &n_in, &n_out, method0, method1, hard_dc, decompose_dc, qintervals ? q : nullptr,
                            latencies ? l : nullptr, adder_size, carry_size, search_all_decompose_dc, &r);
    return rc == DA_OK ? r : nullptr;
}

static std::atomic<int> g_comm_aborted{0};
void da_comm_abort(void) { g_comm_aborted.store(1); }
static std::atomic<long long> g_shard_elements{0};
int64_t da_shard_exchanged_elements(void) { return (int64_t)g_shard_elements.load(); }

static std::unique_ptr<da::ShardEngine> make_hip_shard(const da::ChainJob &job, int c0, int c1, double capacity_scale, void *ctx) {
    return static_cast<da::gpu::HipBackend *>(ctx)->make_shard_engine(job, c0, c1, capacity_scale);
}

static da_result *solve_sharded_impl(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                                     int decompose_dc, const float *qintervals, const float *latencies, int adder_size, int carry_size,
                                     int search_all_decompose_dc, int rank, int world, da_allreduce_i32 allreduce, void *ctx, int64_t *stats3,
                                     bool stream_ordered);

// the RCCL transport as a ShardComm collective: ctx = the transport; a failed collective raises the abort flag (ShardComm::sum throws)
static void rccl_allreduce_cb(void *ctx, void *buf, int64_t count, int on_device) {
    if (!static_cast<da::gpu::RcclTransport *>(ctx)->allreduce(buf, count, on_device != 0)) g_comm_aborted.store(1);
}

int da_rccl_unique_id(void *id128) {
    try {
        if (!id128) throw std::invalid_argument("id128 must point to 128 bytes");
        da::gpu::rccl_unique_id(id128);
        return DA_OK;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int da_rccl_shutdown(void) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    try {
        return da::gpu::rccl_shutdown();
    } catch (const std::exception &e) {
        fail(e);
        return -1;
    }
}

da_result *da_solve_sharded_rccl(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                                 int decompose_dc, const float *qintervals, const float *latencies, int adder_size, int carry_size,
                                 int search_all_decompose_dc, int rank, int world, const void *id128, int64_t *stats3) {
    std::shared_ptr<da::gpu::RcclTransport> tr;
    {
        std::lock_guard<LibraryMutex> lk(g_mutex);
        try {
            if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("rank must be in [0, world)");
            if (!id128) throw std::invalid_argument("id128 (the 128-byte RCCL unique id of rank 0) must be given");
            da::gpu::HipBackend &inner = backend();
            tr = da::gpu::RcclTransport::open(id128, rank, world, g_device, inner.stream());
        } catch (const std::exception &e) {
            fail(e);
            return nullptr;
        }
    }
    da_result *r = solve_sharded_impl(kernel, n_in, n_out, method0, method1, hard_dc, decompose_dc, qintervals, latencies, adder_size, carry_size,
                                      search_all_decompose_dc, rank, world, rccl_allreduce_cb, tr.get(), stats3, true);
    if (!r && tr->last_error()[0]) g_err += std::string(" [") + tr->last_error() + "]";
    return r;
}

// stream_ordered: the collective is queued on the backend's own stream (the RCCL transport), so the engine hands over device
// buffers without completing its kernels first
static da_result *solve_sharded_impl(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                                     int decompose_dc, const float *qintervals, const float *latencies, int adder_size, int carry_size,
                                     int search_all_decompose_dc, int rank, int world, da_allreduce_i32 allreduce, void *ctx, int64_t *stats3,
                                     bool stream_ordered) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    try {
        if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("rank must be in [0, world)");
        if (world > 255) throw std::invalid_argument("at most 255 ranks (8-bit flag fields)");
        if (world > 1 && !allreduce) throw std::invalid_argument("an all-reduce callback must be given for world > 1");
        if (n_in <= 0 || n_out <= 0) throw std::invalid_argument("kernel must be a non-empty 2-D matrix");
        da::Problem p;
        p.kernel = kernel;
        p.n_in = (int)n_in;
        p.n_out = (int)n_out;
        p.opt.method0 = method0;
        p.opt.method1 = method1;
        p.opt.hard_dc = hard_dc;
        p.opt.decompose_dc = decompose_dc;
        check_dyadic_steps(qintervals, p.n_in);
        if (qintervals)
            for (int r = 0; r < p.n_in; ++r) p.opt.qints.push_back(da::QInt{qintervals[3 * r], qintervals[3 * r + 1], qintervals[3 * r + 2]});
        if (latencies) p.opt.lats.assign(latencies, latencies + p.n_in);
        p.opt.adder_size = adder_size;
        p.opt.carry_size = carry_size;
        p.opt.search_all = search_all_decompose_dc != 0;
        da::gpu::HipBackend &inner = backend();
        da::ShardComm comm;
        comm.rank = rank;
        comm.world = world;
        comm.allreduce = allreduce;
        comm.ctx = ctx;
        g_comm_aborted.store(0);
        comm.aborted = &g_comm_aborted;
        comm.stream_ordered = stream_ordered || std::getenv("DA4ML_SHARD_STREAM_ORDERED") != nullptr;  // (the variable: test hook for transports that do nothing)
        da::ShardedBackend be(inner, comm, make_hip_shard, &inner);
        be.force_single = std::getenv("DA4ML_SHARD_FORCE") != nullptr;
        be.force_comm(std::getenv("DA4ML_SHARD_FORCE_COMM") != nullptr);  // call the collective with one rank too (measurement of the transport)
        std::vector<da::ChainStats> stats;
        std::vector<da::PipeResult> res = da::solve_batch(be, {p}, &stats);
        if (stats3) {
            stats3[0] = be.sharded_chains;
            stats3[1] = be.sharded_steps;
            stats3[2] = be.comm().calls;
        }
        g_shard_elements.store(be.comm().elements);
        return new da_result{std::move(res[0]), stats.empty() ? da::ChainStats{} : stats[0]};
    } catch (const std::exception &e) {
        fail(e);
        return nullptr;
    }
}

da_result *da_solve_sharded(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                            int decompose_dc, const float *qintervals, const float *latencies, int adder_size, int carry_size,
                            int search_all_decompose_dc, int rank, int world, da_allreduce_i32 allreduce, void *ctx, int64_t *stats3) {
    return solve_sharded_impl(kernel, n_in, n_out, method0, method1, hard_dc, decompose_dc, qintervals, latencies, adder_size, carry_size,
                              search_all_decompose_dc, rank, world, allreduce, ctx, stats3, false);
}

int da_n_stages(const da_result *r) { return (int)r->pipe.stages.size(); }
int da_picked(const da_result *r) { return r->pipe.picked; }
int da_stage_info(const da_result *r, int stage, int64_t *info) {
    const da::StageResult &s = r->pipe.stages[stage];
    info[0] = s.n_in;
    info[1] = s.n_out;
    info[2] = (int64_t)s.ops.size();
    info[3] = s.carry_size;
    info[4] = s.adder_size;
    return DA_OK;
}
int da_stage_copy(const da_result *r, int stage, int64_t *inp_shifts, int64_t *out_idxs, int64_t *out_shifts, int64_t *out_negs,
                  int64_t *ops_i, float *ops_f) {
    const da::StageResult &s = r->pipe.stages[stage];
    std::memcpy(inp_shifts, s.inp_shifts.data(), s.inp_shifts.size() * 8);
    std::memcpy(out_idxs, s.out_idxs.data(), s.out_idxs.size() * 8);
    std::memcpy(out_shifts, s.out_shifts.data(), s.out_shifts.size() * 8);
    std::memcpy(out_negs, s.out_negs.data(), s.out_negs.size() * 8);
    for (size_t k = 0; k < s.ops.size(); ++k) {
        const da::OpRec &o = s.ops[k];
        int64_t *oi = ops_i + 4 * k;
        float *of = ops_f + 5 * k;
        oi[0] = o.id0, oi[1] = o.id1, oi[2] = o.opcode, oi[3] = o.data;
        of[0] = o.q.lo, of[1] = o.q.hi, of[2] = o.q.step, of[3] = o.latency, of[4] = o.cost;
    }
    return DA_OK;
}
int da_result_stats(const da_result *r, int64_t *s) {
    const da::ChainStats &t = r->stats;
    int64_t v[8] = {t.iterations, t.digits0, t.blocks0, t.rebuilds, t.table_peak, t.scan_slots, t.partners, t.matches};
    std::memcpy(s, v, sizeof v);
    return DA_OK;
}
void da_free(da_result *r) {
    if (!r) return;
    for (da::StageResult &st : r->pipe.stages) da::recycle_op_list(std::move(st.ops));  // kept for the next call (cmvm_host.h)
    delete r;
}

int da_timings(double *t, int reset) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    try {
        da::gpu::HipBackend &be = backend();
        const da::gpu::GpuTimings &g = be.timings();
        double v[32] = {g.loop_ms,          g.dist_ms,         g.total_ms,         (double)g.lockstep_iters, (double)g.iterations,
                        (double)g.rescans, (double)g.partners, (double)g.chains, g.table_bytes,            g.arena_bytes,
                        g.select_ms_sampled, g.update_ms_sampled, (double)g.samples, (double)g.found, (double)g.inserts,
                        (double)g.cell_reads, g.key_bytes, g.cell_bytes};
        for (int q = 0; q < 12; ++q) v[18 + q] = g.phase_cycles[q];
        v[30] = (double)g.retries;
        v[31] = g.sampled_chain_launches;
        std::memcpy(t, v, sizeof v);
        if (reset) be.reset_timings();
        return DA_OK;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int da_engine_stats(double *out, int n) {
    std::lock_guard<LibraryMutex> lk(g_mutex);
    try {
        const da::gpu::GpuTimings &g = backend().timings();
        const double v[16] = {g.select_bytes, g.host_launch_ms, (double)g.fast_steps, g.search_cycles[0], g.search_cycles[1], g.search_cycles[2], g.search_cycles[3], g.search_diag[0], g.search_diag[1], g.search_diag[2], g.search_diag[3], g.search_diag[4], g.search_diag[5], g.search_diag[6], 0.0, 0.0};
        const int m = n < 16 ? (n < 0 ? 0 : n) : 16;
        for (int i = 0; i < m; ++i) out[i] = v[i];
        return m;
    } catch (const std::exception &e) {
        fail(e);
        return 0;
    }
}

}  // extern "C"
