// cmvm_shard.h -- one greedy chain sharded over the output COLUMNS of its matrix (BASELINE config C4: "256x256 int8
// matrix, 8 x MI355X column-sharded search with RCCL allreduce of shared-subexpression costs").
//
// Every counting loop of the reference has the output column outermost (state_opr.cc:117 initial pair counts, :249
// substitution, :307 recount) and the adder trees are per column (cmvm_core.cc:103): pair counts are SUMS over columns.
// Rank g of W therefore owns the columns [g n_out / W, (g+1) n_out / W) of every row (digits, column lists) while the
// pair table -- counts, scores, the arg-max -- is REPLICATED: every rank holds all of it and takes the identical,
// deterministic decision without an exchange.  Per greedy step the ranks exchange
//   (1) which rows share a substituted column with the consumed digits on SOME rank (flags, all-reduce(sum)), and
//   (2) the partial count changes of those rows' blocks over the own columns (one slab, all-reduce(sum)),
// then every rank applies the summed slab to its copy of the table.  One more all-reduce at the start (initial pair
// counts) and three at the end (surviving digits of all columns, so that every rank can run the adder trees and
// return the complete result).  The only collective is all-reduce(sum) of int32 -- RCCL over xGMI on GPUs.
//
// Two collectives per greedy step, ~2 10^4 steps per 256x256 chain, each bound by link latency, not bandwidth: this
// layout cannot scale near-linearly (SURVEY.md section 8e says so); it is built and measured because the configuration
// names it.  The instance-sharded layout (multi_gpu.solve_many_sharded) is the one that scales.
#pragma once

#include <atomic>
#include <cstdint>
#include <memory>
#include <stdexcept>

#include "cmvm_host.h"

namespace da {

// all-reduce(sum) over `count` int32 values at `buf`, in place, over all ranks.  `on_device` tells whether `buf` is
// device memory (the HIP engine's slabs) or host memory (engine model, host-side merge).  The caller guarantees that
// the producing work is complete (stream synchronised); the callee returns when the result is in place.
typedef void (*allreduce_i32_fn)(void *ctx, void *buf, int64_t count, int on_device);

struct ShardComm {
    int rank = 0, world = 1;
    allreduce_i32_fn allreduce = nullptr;
    void *ctx = nullptr;
    long long calls = 0, elements = 0;  // statistics
    // set by the callback's owner when a collective failed (the callback itself returns nothing and must not unwind through
    // this code): the chain stops at once instead of going on with a buffer that was not reduced
    std::atomic<int> *aborted = nullptr;
    bool stream_ordered = false;  // device buffers are reduced on the engine's stream (no completed-buffer contract needed)
    bool force = false;  // test / measurement aid: call the collective with a single rank too (an all-reduce over one rank)
    void sum(void *buf, int64_t count, bool on_device) {
        if ((world > 1 || force) && allreduce && count > 0) allreduce(ctx, buf, count, on_device ? 1 : 0);
        if (aborted && aborted->exchange(0)) throw std::runtime_error("the all-reduce callback reported a failed collective");
        ++calls;
        elements += count;
    }
};

inline void shard_columns(int n_out, int rank, int world, int &c0, int &c1) {
    c0 = (int)((long long)n_out * rank / world);
    c1 = (int)((long long)n_out * (rank + 1) / world);
}
// 8-bit fields of the flag words: four rows per int32, summed over <= 255 ranks without carry between fields
inline int64_t flag_words(int n_rows) { return (n_rows + 3) / 4; }
// The flag buffer of a step ends in a STATUS TRAILER of three words, summed over the ranks with the flags -- no extra collective:
// {ranks that have stopped, of which with a capacity error (row / table / list arena), of which with any other error}.  Every
// rank takes part in the flag exchange of every step, also one that has stopped (all flags zero): a rank-local error can
// therefore not leave the others waiting in a collective, and every rank takes the same decision (stop, or retry with larger
// arenas) from the same summed words.
constexpr int SHARD_TRAILER = 3;
inline bool is_capacity_error(int e) { return e == E_ROW_CAPACITY || e == E_TABLE_CAPACITY || e == E_LIST_CAPACITY; }

// One chain on one rank's columns.  Buffers handed out by the engine live where `on_device()` says.
class ShardEngine {
  public:
    virtual ~ShardEngine() = default;
    virtual bool on_device() const = 0;
    virtual int n_keys() const = 0;  // keys per pair block
    // partial counts of all pairs of input rows (i0 <= i1, index i1 (i1+1)/2 + i0) over the own columns: int32 [n_pairs][K]
    virtual int32_t *init_counts(int64_t &count) = 0;
    virtual void init_table() = 0;  // from the (summed) buffer returned by init_counts
    // phase 1 of a greedy step: arg-max on the replicated table, substitution in the own columns.  ALWAYS hands out the flag
    // buffer of the step: flag_words(rows before the step) words of packed 8-bit fields (field r != 0 <=> row r shares a
    // substituted column here) followed by the status trailer; an engine that has finished or failed hands out zero flags and
    // the trailer {1, capacity error?, other error?}.
    virtual void select(int32_t *&flags, int64_t &flag_count) = 0;
    // phase 2: from the summed flags, the union of partner rows (ascending ids, identical on every rank) and this
    // rank's partial count changes: slab int32 [(6 + 3 n_union)][K] = the six pairs among {A, B, new} (AA, AB, BB, AN, BN,
    // NN), then per partner {lost with A, lost with B, gained with the new row}.  status = the summed trailer; when
    // status[0] != 0 (somebody has stopped) nothing is computed and nullptr is returned.
    virtual int32_t *partial(int64_t &slab_count, int32_t status[SHARD_TRAILER]) = 0;
    virtual void apply() = 0;  // phase 3: the summed slab into the table
    // the collectives are queued on the engine's own stream (the library's RCCL transport): the engine need not complete its
    // kernels before handing a device buffer to ShardComm::sum
    virtual void set_stream_ordered(bool) {}
    // own columns of the finished chain (col_start covers the own columns only; everything row-related is global)
    virtual void finish(ChainOut &own) = 0;
};

// capacity_scale multiplies the engine's arena heuristics (1 on the first attempt, x 4 per retry after a capacity error)
using ShardEngineFactory = std::unique_ptr<ShardEngine> (*)(const ChainJob &job, int c0, int c1, double capacity_scale, void *factory_ctx);

// Backend whose chains are column-sharded; everything else (stage-1 distances, decompositions, chains that cannot be
// sharded: fewer columns than ranks, the "dummy" method) goes to `inner`, replicated on every rank.
class ShardedBackend : public Backend {
  public:
    ShardedBackend(Backend &inner, ShardComm comm, ShardEngineFactory make, void *factory_ctx)
        : inner_(inner), comm_(comm), make_(make), ctx_(factory_ctx) {}
    void run_chains(const ChainJob *jobs, ChainOut *outs, int n) override;
    void column_distances(const int32_t *aug, int n_in, int W, int64_t *d0, int64_t *d1) override { inner_.column_distances(aug, n_in, W, d0, d1); }
    int csd_decompose(const float *kernel, int n_in, int n_out, bool center, std::vector<int8_t> &csd, std::vector<int8_t> &s0,
                      std::vector<int8_t> &s1) override {
        return inner_.csd_decompose(kernel, n_in, n_out, center, csd, s0, s1);
    }
    int int_to_csd(const int32_t *x, int64_t n, std::vector<int8_t> &csd) override { return inner_.int_to_csd(x, n, csd); }
    const ShardComm &comm() const { return comm_; }
    void force_comm(bool on) { comm_.force = on; }
    long long sharded_chains = 0, sharded_steps = 0, capacity_retries = 0;
    bool force_single = false;  // test aid: run the sharded phases with a single rank too (the exchanges are no-ops)

  private:
    void run_one(const ChainJob &job, ChainOut &out);
    Backend &inner_;
    ShardComm comm_;
    ShardEngineFactory make_;
    void *ctx_;
};

}  // namespace da
