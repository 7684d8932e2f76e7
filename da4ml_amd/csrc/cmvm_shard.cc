// cmvm_shard.cc -- orchestration of a column-sharded greedy chain (see cmvm_shard.h).  Engine-agnostic: the HIP engine
// (cmvm_shard_gpu.hip) in the product, the sequential engine model in the CPU tests.

#include "cmvm_shard.h"

#include <algorithm>
#include <numeric>

namespace da {

void ShardedBackend::run_chains(const ChainJob *jobs, ChainOut *outs, int n) {
    for (int i = 0; i < n; ++i) {
        const ChainJob &j = jobs[i];
        const bool shardable = (comm_.world > 1 || force_single) && j.n_out >= comm_.world && j.method >= 0 && j.method != M_DUMMY;
        if (shardable)
            run_one(j, outs[i]);
        else
            inner_.run_chains(&j, &outs[i], 1);  // replicated: every rank computes the same small chain
    }
}

void ShardedBackend::run_one(const ChainJob &job, ChainOut &out) {
    int c0, c1;
    shard_columns(job.n_out, comm_.rank, comm_.world, c0, c1);
    ++sharded_chains;
    std::unique_ptr<ShardEngine> eng;
    int32_t status[SHARD_TRAILER] = {0, 0, 0};
    double scale = 1.0;
    for (int attempt = 0;; ++attempt) {
        // the arena sizes are heuristics (as in HipBackend::run_chains): a capacity error reruns the chain with four times the
        // capacities, at most three times.  The decision comes from the SUMMED status words, so every rank takes it together.
        eng = make_(job, c0, c1, scale, ctx_);
        const bool dev = eng->on_device();
        eng->set_stream_ordered(comm_.stream_ordered);

        // initial pair counts: partial over the own columns, summed over the ranks
        int64_t count = 0;
        int32_t *buf = eng->init_counts(count);
        comm_.sum(buf, count, dev);
        eng->init_table();

        // greedy loop: two exchanges per step; the first one carries every rank's status (SHARD_TRAILER)
        while (true) {
            int32_t *flags = nullptr;
            int64_t fcount = 0;
            eng->select(flags, fcount);
            comm_.sum(flags, fcount, dev);
            int64_t scount = 0;
            int32_t *slab = eng->partial(scount, status);
            if (status[0] != 0) break;  // the chain is finished (the table is replicated: on every rank in the same step) or somebody failed
            comm_.sum(slab, scount, dev);
            eng->apply();
            ++sharded_steps;
        }
        if (status[1] != 0 && status[2] == 0 && attempt < 3) {
            scale *= 4.0;
            ++capacity_retries;
            continue;
        }
        break;
    }
    const bool failed = status[1] != 0 || status[2] != 0 || status[0] != comm_.world;

    // merge: everything row-related is already global; the surviving digits of the other ranks' columns come in by
    // all-reduce(sum) of buffers in which every rank fills its own segment
    ChainOut own;
    eng->finish(own);
    out = ChainOut{};
    if (failed) {  // known to every rank from the same summed words: nobody enters the merge exchanges
        // this rank's own error if it has one; else what the summed status says about the others: a rank with a non-capacity error, or ranks
        // out of step (fewer than `world` stopped together) -> E_REMOTE_ERROR, which no caller retries; only a pure capacity failure (the
        // retries above are used up) is reported as one
        // (decided on the error words first: a pure capacity failure stops only the failing rank, so status[0] != world there too -- every
        // rank reports it as a capacity error then; "out of step" is what is left when no rank raised an error flag)
        out.error = own.error ? own.error : status[2] != 0 ? E_REMOTE_ERROR : status[1] != 0 ? E_TABLE_CAPACITY : E_REMOTE_ERROR;
        out.n_bits = own.n_bits;
        return;
    }
    out.error = own.error;
    out.unknown_method_hit = own.unknown_method_hit;
    out.n_bits = own.n_bits;
    out.shift0 = own.shift0;
    out.shift1 = own.shift1;
    out.picks = own.picks;
    out.row_lat = own.row_lat;
    out.stats = own.stats;
    const int n_out = job.n_out;
    std::vector<int32_t> cnt((size_t)n_out + 1, 0);
    for (int j = c0; j < c1; ++j) cnt[j] = (int32_t)(own.col_start[j - c0 + 1] - own.col_start[j - c0]);
    cnt[n_out] = own.error;  // any rank's capacity error fails the chain everywhere
    comm_.sum(cnt.data(), (int64_t)cnt.size(), false);
    if (cnt[n_out] != 0 && out.error == E_OK) out.error = own.error ? own.error : E_TABLE_CAPACITY;
    out.col_start.assign((size_t)n_out + 1, 0);
    for (int j = 0; j < n_out; ++j) out.col_start[j + 1] = out.col_start[j] + (uint32_t)cnt[j];
    const size_t total = out.col_start[n_out];
    std::vector<int32_t> dig(3 * total, 0);  // row | low half of the cell | high half
    const size_t first = out.col_start[c0];
    for (size_t k = 0; k < own.dig_row.size(); ++k) {
        dig[first + k] = (int32_t)own.dig_row[k];
        dig[total + first + k] = (int32_t)(uint32_t)own.dig_cell[k];
        dig[2 * total + first + k] = (int32_t)(uint32_t)(own.dig_cell[k] >> 32);
    }
    comm_.sum(dig.data(), (int64_t)dig.size(), false);
    out.dig_row.resize(total);
    out.dig_cell.resize(total);
    for (size_t k = 0; k < total; ++k) {
        out.dig_row[k] = (uint32_t)dig[k];
        out.dig_cell[k] = (uint64_t)(uint32_t)dig[total + k] | ((uint64_t)(uint32_t)dig[2 * total + k] << 32);
    }
    // statistics that are per-column sums
    int32_t st[2] = {(int32_t)std::min<int64_t>(own.stats.matches, INT32_MAX), (int32_t)std::min<int64_t>(own.stats.digits0, INT32_MAX)};
    comm_.sum(st, 2, false);
    out.stats.matches = st[0];
    out.stats.digits0 = st[1];
}

}  // namespace da
