// cmvm_gpu.h -- the HIP implementation of da::Backend (kernels in cmvm_engine.hip).
#pragma once

#include <memory>

#include "cmvm_host.h"
#include "cmvm_shard.h"

namespace da {
namespace gpu {

struct GpuTimings {  // accumulated since the last reset; read by the benchmark harness
    double loop_ms = 0;       // HIP-event time of the greedy loops (k_iter_select + k_iter_update launches)
    double dist_ms = 0;       // HIP-event time of k_col_dist
    double total_ms = 0;      // wall time inside run_chains (uploads, set-up, loop, downloads)
    long long lockstep_iters = 0;  // launched (select, update) kernel pairs
    long long iterations = 0;      // greedy iterations summed over chains
    long long rescans = 0;         // table groups re-read by the selection
    long long partners = 0;        // partner rows processed by the update kernel
    long long chains = 0;
    long long retries = 0;         // capacity retries (arena heuristics too small)
    long long dist_calls = 0;
    // sampled per-kernel durations (HIP events around every 16th lockstep iteration)
    double select_ms_sampled = 0, update_ms_sampled = 0;
    long long samples = 0;
    double sampled_chain_launches = 0;  // sum over sampled launches of the number of chains in that launch
    // algorithmic-traffic counters of k_iter_update, summed over chains
    long long found = 0, inserts = 0, cell_reads = 0;
    double key_bytes = 0;     // 2 * K bytes per touched count block (u16 counts)
    double cell_bytes = 0;    // bytes of partner cells read
    double phase_cycles[12] = {0};  // shader-clock cycles: k_iter_select phases 0-6, k_iter_update per-wave phases 7-11
    double table_bytes = 0;   // bytes of pair-table storage summed over chains
    double arena_bytes = 0;   // largest device arena used
    double select_bytes = 0;    // algorithmic bytes of k_iter_select, counted on the device
    double host_launch_ms = 0;  // host time spent queueing the greedy loop's launches (the launch thread's share of the loop)
    long long fast_steps = 0;   // greedy steps whose pick was known before the step began (k_iter_select2)
    double search_diag[7] = {0};    // (phase-timer builds) re-reads of stale groups below the floor, of clean groups with an excluded best entry, rounds of the longest wave
    double search_cycles[4] = {0};  // shader-clock cycles of the search block: bounds, arg-max of steps without a known pick, search; [3] = steps timed
};

class HipBackend : public Backend {
  public:
    explicit HipBackend(int device = 0);
    ~HipBackend() override;
    void run_chains(const ChainJob *jobs, ChainOut *outs, int n) override;
    void column_distances(const int32_t *aug, int n_in, int W, int64_t *d0, int64_t *d1) override;
    int csd_decompose(const float *kernel, int n_in, int n_out, bool center, std::vector<int8_t> &csd, std::vector<int8_t> &s0,
                      std::vector<int8_t> &s1) override;
    int int_to_csd(const int32_t *x, int64_t n, std::vector<int8_t> &csd) override;

    const GpuTimings &timings() const;
    void reset_timings();
    void *stream() const;  // hipStream_t the kernels are launched on
    // one greedy chain on the columns [c0, c1) of its matrix, pair table replicated (cmvm_shard.h); the engine lives on
    // this backend's device and stream and must not outlive it
    std::unique_ptr<ShardEngine> make_shard_engine(const ChainJob &job, int c0, int c1, double capacity_scale = 1.0);

  private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    double row_scale_ = 1.0;
    int retry_depth_ = 0;
};

int device_count();

}  // namespace gpu
}  // namespace da
