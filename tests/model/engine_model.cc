// engine_model.cc -- TEST INFRASTRUCTURE: a sequential, single-threaded model of the GPU greedy engine.
//
// It implements da::Backend with exactly the data model and update rules of da4ml_amd/csrc/cmvm_engine.hip
// (cells as position bitmasks, one count block per row pair, incremental "delta" maintenance of the pair
// counts instead of the reference's purge + regenerate, rank/tie-word selection) using the same inline
// functions from cmvm_core.h, but with plain loops and an std::unordered_map instead of kernels and the
// device hash table.  It is linked with the product's host logic (cmvm_host.cc) into tests/model/libmodel.so
// so that `pytest -m "not gpu"` can prove, without a GPU, that
//   (1) the reformulated algorithm picks exactly the reference's pairs, and
//   (2) the host orchestration / adder-tree code reproduces the reference's op lists.
// It is never linked into the product library.

#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../da4ml_amd/csrc/cmvm_core.h"
#include "../../da4ml_amd/csrc/cmvm_host.h"
#include "../../da4ml_amd/csrc/cmvm_shard.h"

namespace {
using namespace da;

struct Block {
    std::vector<uint16_t> cnt;
    int ov;
    float dl;
    uint32_t rank = 0;
    int best = -1;
};

template <class Cell> struct Chain {
    using O = CellOps<Cell>;
    int n_in, n_out, N, K, method, adder, carry;
    std::vector<std::vector<Cell>> cells;  // [row][col]
    std::vector<RowInfo> rows;
    std::vector<std::vector<uint32_t>> collist;  // rows that have (had) digits in a column
    std::unordered_map<uint64_t, Block> table;   // key = id1 << 32 | id0
    Log2Table tab;
    StepLog2Host step_tab;  // -log2f of non-power-of-two input steps, as the product's host builds it
    int err = 0;
    ChainStats st;

    static uint64_t pkey(uint32_t a, uint32_t b) { return ((uint64_t)b << 32) | a; }

    void refresh_best(Block &b) {
        b.rank = 0;
        b.best = -1;
        for (int k = 0; k < K; ++k) {
            uint32_t r = entry_rank(b.cnt[k], b.ov, b.dl, method);
            if (r != 0 && r >= b.rank) {
                b.rank = r;
                b.best = k;
            }
        }
    }
    bool alive(const Block &b) const {
        for (int k = 0; k < K; ++k)
            if (b.cnt[k] >= 2) return true;
        return false;
    }
    Block fresh(uint32_t lo, uint32_t hi) {
        Block b;
        b.cnt.assign(K, 0);
        b.ov = n_overlap(rows[lo], rows[hi]);
        b.dl = __builtin_fabsf(rows[lo].lat - rows[hi].lat);
        return b;
    }
    // exact recount of one row pair over all columns
    void recount(uint32_t lo, uint32_t hi) {
        Block b = fresh(lo, hi);
        for (int j = 0; j < n_out; ++j) {
            if (lo == hi)
                for_pairs_self<Cell>(cells[lo][j], N, [&](int k) { b.cnt[k]++; });
            else
                for_pairs_cross<Cell>(cells[lo][j], cells[hi][j], N, [&](int k) { b.cnt[k]++; });
        }
        uint64_t key = pkey(lo, hi);
        if (alive(b)) {
            refresh_best(b);
            table[key] = std::move(b);
        } else
            table.erase(key);
    }

    void init(const ChainJob &job, ChainOut &out) {
        n_in = job.n_in;
        n_out = job.n_out;
        method = job.method;
        adder = job.adder_size;
        carry = job.carry_size;
        tab = measure_log2_table();
        step_tab.build(job.qints, job.n_in);
        std::vector<float> a(job.kernel, job.kernel + (size_t)n_in * n_out);
        center_matrix(a, n_in, n_out, out.shift0, out.shift1);
        uint32_t mx = 0;
        for (float v : a) mx = std::max(mx, (uint32_t)std::abs((int32_t)v));
        N = csd_width(mx);
        out.n_bits = N;
        K = key_count(N);
        cells.assign(n_in, std::vector<Cell>(n_out, 0));
        collist.assign(n_out, {});
        rows.resize(n_in);
        for (int i = 0; i < n_in; ++i) {
            rows[i] = RowInfo{job.qints[i].lo, job.qints[i].hi, job.qints[i].step, job.lats[i]};
            bool dead = job.qints[i].lo == 0.0f && job.qints[i].hi == 0.0f;
            for (int j = 0; j < n_out; ++j) {
                uint32_t p, m;
                naf_masks((int32_t)a[(size_t)i * n_out + j], p, m);
                if (dead) p = m = 0;
                cells[i][j] = O::make(p, m);
                if (p | m) {
                    collist[j].push_back(i);
                    st.digits0 += popc32(p | m);
                }
            }
        }
        if (method == M_DUMMY) return;
        for (int i0 = 0; i0 < n_in; ++i0)
            for (int i1 = i0; i1 < n_in; ++i1) recount(i0, i1);
        st.blocks0 = (int64_t)table.size();
    }

    bool select(uint32_t &A, uint32_t &B, int &idx) {
        uint32_t best_rank = 0;
        uint64_t best_tie = 0;
        st.scan_slots += (int64_t)table.size();
        for (auto &kv : table) {
            const Block &b = kv.second;
            if (b.rank == 0) continue;
            uint32_t id0 = (uint32_t)kv.first, id1 = (uint32_t)(kv.first >> 32);
            uint64_t tw = tie_word(id0, id1, b.best);
            if (b.rank > best_rank || (b.rank == best_rank && tw > best_tie)) {
                best_rank = b.rank;
                best_tie = tw;
                A = id0;
                B = id1;
                idx = b.best;
            }
        }
        return best_rank != 0;
    }

    void step(uint32_t A, uint32_t B, int idx, ChainOut &out) {
        int shift, sub;
        key_decode(idx, N, shift, sub);
        uint32_t Nw = (uint32_t)cells.size();
        // new row record
        RowInfo ni;
        qint_add_pair(rows[A], rows[B], shift, sub, ni.lo, ni.hi, ni.step);
        float dlat = adder_dlat(rows[A], rows[B], shift, sub, adder, carry, tab, step_tab.view(), err);
        ni.lat = (rows[A].lat < rows[B].lat ? rows[B].lat : rows[A].lat) + dlat;
        rows.push_back(ni);
        out.picks.insert(out.picks.end(), {(int32_t)A, (int32_t)B, sub, shift});
        // substitution
        cells.emplace_back(n_out, 0);
        std::vector<int> mcol;
        std::vector<Cell> MA, MB;
        for (int j = 0; j < n_out; ++j) {
            Cell ma, mb;
            substitute_column<Cell>(cells[A][j], cells[B][j], A == B, shift, sub, ma, mb);
            if (!ma) continue;
            cells[A][j] &= ~ma;
            cells[B][j] &= ~mb;
            cells[Nw][j] = ma;
            mcol.push_back(j);
            MA.push_back(ma);
            MB.push_back(mb);
            st.matches += popc32(O::plus(ma) | O::minus(ma));
        }
        // pairs among the modified rows: exact recount
        recount(A, A);
        if (A != B) {
            recount(A, B);
            recount(B, B);
            recount(B, Nw);
        }
        recount(A, Nw);
        recount(Nw, Nw);
        // partners: every other row with digits in a matched column
        std::map<uint32_t, int> seen;
        for (int j : mcol)
            for (uint32_t r : collist[j])
                if (r != A && r != B) seen.emplace(r, 0);
        for (int j : mcol) collist[j].push_back(Nw);
        std::vector<int> dA(K), dB(K), cN(K);
        for (auto &pr : seen) {
            uint32_t r = pr.first;
            st.partners++;
            std::fill(dA.begin(), dA.end(), 0);
            std::fill(dB.begin(), dB.end(), 0);
            std::fill(cN.begin(), cN.end(), 0);
            for (size_t q = 0; q < mcol.size(); ++q) {
                Cell x = cells[r][mcol[q]];
                if (!x) continue;
                for_pairs_part<Cell>(MA[q], x, A < r, N, [&](int k) { dA[k]++; });
                if (A != B)
                    for_pairs_part<Cell>(MB[q], x, B < r, N, [&](int k) { dB[k]++; });
                else
                    for_pairs_part<Cell>(MB[q], x, A < r, N, [&](int k) { dA[k]++; });
                for_pairs_cross<Cell>(x, MA[q], N, [&](int k) { cN[k]++; });
            }
            auto apply = [&](uint32_t m, std::vector<int> &d) {
                auto it = table.find(pkey(std::min(m, r), std::max(m, r)));
                if (it == table.end()) return;
                Block &b = it->second;
                for (int k = 0; k < K; ++k) b.cnt[k] = (uint16_t)(b.cnt[k] - d[k]);
                if (alive(b))
                    refresh_best(b);
                else
                    table.erase(it);
            };
            apply(A, dA);
            if (A != B) apply(B, dB);
            Block nb = fresh(r, Nw);
            bool any = false;
            for (int k = 0; k < K; ++k) {
                nb.cnt[k] = (uint16_t)cN[k];
                any |= cN[k] >= 2;
            }
            if (any) {
                refresh_best(nb);
                table[pkey(r, Nw)] = std::move(nb);
            }
        }
        st.table_peak = std::max(st.table_peak, (int64_t)table.size());
    }

    void run(const ChainJob &job, ChainOut &out) {
        init(job, out);
        if (method == -1 && !table.empty()) out.unknown_method_hit = true;
        if (method >= 0 && method != M_DUMMY) {
            uint32_t A, B;
            int idx;
            while (select(A, B, idx)) {
                step(A, B, idx, out);
                st.iterations++;
            }
        }
        out.error = err ? E_FLOAT_DOMAIN : E_OK;
        for (auto &r : rows) out.row_lat.push_back(r.lat);
        out.col_start.assign(n_out + 1, 0);
        for (int j = 0; j < n_out; ++j) {
            for (size_t r = 0; r < cells.size(); ++r)
                if (cells[r][j]) {
                    out.dig_row.push_back((uint32_t)r);
                    out.dig_cell.push_back((uint64_t)O::plus(cells[r][j]) | ((uint64_t)O::minus(cells[r][j]) << 32));
                }
            out.col_start[j + 1] = (uint32_t)out.dig_row.size();
        }
        out.stats = st;
    }
};

// Column-sharded form of the chain (da::ShardEngine, cmvm_shard.h): the digits of the own columns only, the pair table
// replicated.  Sequential model of what cmvm_shard_gpu.hip does with kernels; lets the CPU tests run the product's
// sharded orchestration (cmvm_shard.cc) over a gloo group.
template <class Cell> struct ShardChain : Chain<Cell>, ShardEngine {
    using Base = Chain<Cell>;
    using O = CellOps<Cell>;
    using Base::cells; using Base::collist; using Base::K; using Base::N; using Base::rows; using Base::table; using Base::st;  // clang-format off
    int c0, c1, n_loc, n_in_;
    ChainOut head;  // global parts: shifts, picks
    std::vector<int32_t> buf, flags, slab;
    uint32_t A = 0, B = 0, Nw = 0;
    std::vector<int> mcol;
    std::vector<Cell> MA, MB;
    std::vector<uint32_t> uni;  // union of partner rows of the current step

    ShardChain(const ChainJob &job, int c0_, int c1_) : c0(c0_), c1(c1_), n_loc(c1_ - c0_), n_in_(job.n_in) {
        this->n_in = job.n_in;
        this->n_out = n_loc;  // the base class loops over the columns held here
        this->method = job.method;
        this->adder = job.adder_size;
        this->carry = job.carry_size;
        this->tab = measure_log2_table();
        this->step_tab.build(job.qints, job.n_in);
        std::vector<float> a(job.kernel, job.kernel + (size_t)job.n_in * job.n_out);
        center_matrix(a, job.n_in, job.n_out, head.shift0, head.shift1);  // centring and digit width are properties of the WHOLE matrix
        uint32_t mx = 0;
        for (float v : a) mx = std::max(mx, (uint32_t)std::abs((int32_t)v));
        N = csd_width(mx);
        head.n_bits = N;
        K = key_count(N);
        cells.assign(job.n_in, std::vector<Cell>(n_loc, 0));
        collist.assign(n_loc, {});
        rows.resize(job.n_in);
        for (int i = 0; i < job.n_in; ++i) {
            rows[i] = RowInfo{job.qints[i].lo, job.qints[i].hi, job.qints[i].step, job.lats[i]};
            bool dead = job.qints[i].lo == 0.0f && job.qints[i].hi == 0.0f;
            for (int j = 0; j < n_loc; ++j) {
                uint32_t p, m;
                naf_masks((int32_t)a[(size_t)i * job.n_out + c0 + j], p, m);
                if (dead) p = m = 0;
                cells[i][j] = O::make(p, m);
                if (p | m) {
                    collist[j].push_back(i);
                    st.digits0 += popc32(p | m);
                }
            }
        }
    }
    bool on_device() const override { return false; }
    int n_keys() const override { return K; }
    void count_pair(uint32_t lo, uint32_t hi, int32_t *out) {  // partial counts of one row pair over the own columns
        for (int j = 0; j < n_loc; ++j) {
            if (lo == hi)
                for_pairs_self<Cell>(cells[lo][j], N, [&](int k) { out[k]++; });
            else
                for_pairs_cross<Cell>(cells[lo][j], cells[hi][j], N, [&](int k) { out[k]++; });
        }
    }
    void put_block(uint32_t lo, uint32_t hi, const int32_t *cnt) {  // replace the block of a pair by summed counts
        Block b = this->fresh(lo, hi);
        for (int k = 0; k < K; ++k) b.cnt[k] = (uint16_t)cnt[k];
        uint64_t key = Base::pkey(lo, hi);
        if (this->alive(b)) {
            this->refresh_best(b);
            table[key] = std::move(b);
        } else
            table.erase(key);
    }
    int32_t *init_counts(int64_t &count) override {
        const int64_t n_pairs = (int64_t)n_in_ * (n_in_ + 1) / 2;
        buf.assign((size_t)n_pairs * K, 0);
        for (int i1 = 0; i1 < n_in_; ++i1)
            for (int i0 = 0; i0 <= i1; ++i0) count_pair(i0, i1, buf.data() + ((size_t)i1 * (i1 + 1) / 2 + i0) * K);
        count = (int64_t)buf.size();
        return buf.data();
    }
    void init_table() override {
        for (int i1 = 0; i1 < n_in_; ++i1)
            for (int i0 = 0; i0 <= i1; ++i0) put_block(i0, i1, buf.data() + ((size_t)i1 * (i1 + 1) / 2 + i0) * K);
        st.blocks0 = (int64_t)table.size();
    }
    bool stopped = false;
    void select(int32_t *&fl, int64_t &fcount) override {
        int idx;
        if (stopped || !Base::select(A, B, idx)) {  // finished (or failed): zero flags + status trailer, the rank still takes part in the exchange
            stopped = true;
            flags.assign((size_t)flag_words((int)cells.size()) + SHARD_TRAILER, 0);
            flags[flags.size() - 3] = 1;
            flags[flags.size() - 1] = this->err ? 1 : 0;
            fl = flags.data();
            fcount = (int64_t)flags.size();
            return;
        }
        int shift, sub;
        key_decode(idx, N, shift, sub);
        Nw = (uint32_t)cells.size();
        RowInfo ni;
        qint_add_pair(rows[A], rows[B], shift, sub, ni.lo, ni.hi, ni.step);
        float dlat = adder_dlat(rows[A], rows[B], shift, sub, this->adder, this->carry, this->tab, this->step_tab.view(), this->err);
        ni.lat = (rows[A].lat < rows[B].lat ? rows[B].lat : rows[A].lat) + dlat;
        rows.push_back(ni);
        head.picks.insert(head.picks.end(), {(int32_t)A, (int32_t)B, sub, shift});
        cells.emplace_back(n_loc, 0);
        mcol.clear();
        MA.clear();
        MB.clear();
        for (int j = 0; j < n_loc; ++j) {
            Cell ma, mb;
            substitute_column<Cell>(cells[A][j], cells[B][j], A == B, shift, sub, ma, mb);
            if (!ma) continue;
            cells[A][j] &= ~ma;
            cells[B][j] &= ~mb;
            cells[Nw][j] = ma;
            mcol.push_back(j);
            MA.push_back(ma);
            MB.push_back(mb);
            st.matches += popc32(O::plus(ma) | O::minus(ma));
        }
        flags.assign((size_t)flag_words((int)Nw) + SHARD_TRAILER, 0);
        for (int j : mcol)
            for (uint32_t r : collist[j])
                if (r != A && r != B) flags[r >> 2] |= 1 << (8 * (r & 3));
        for (int j : mcol) collist[j].push_back(Nw);
        st.iterations++;
        fl = flags.data();
        fcount = (int64_t)flags.size();
    }
    int32_t *partial(int64_t &scount, int32_t status[SHARD_TRAILER]) override {
        for (int q = 0; q < SHARD_TRAILER; ++q) status[q] = flags[flags.size() - SHARD_TRAILER + q];
        scount = 0;
        if (status[0] != 0) return nullptr;
        uni.clear();
        for (uint32_t r = 0; r < Nw; ++r)
            if ((flags[r >> 2] >> (8 * (r & 3))) & 0xFF) uni.push_back(r);
        const int Kk = K;
        slab.assign((6 + 3 * uni.size()) * (size_t)Kk, 0);
        for (size_t u = 0; u < uni.size(); ++u) {
            uint32_t r = uni[u];
            int32_t *dA = slab.data() + (6 + 3 * u) * Kk, *dB = dA + Kk, *cN = dB + Kk;
            for (size_t q = 0; q < mcol.size(); ++q) {
                Cell x = cells[r][mcol[q]];
                if (!x) continue;
                for_pairs_part<Cell>(MA[q], x, A < r, N, [&](int k) { dA[k]++; });
                if (A != B)
                    for_pairs_part<Cell>(MB[q], x, B < r, N, [&](int k) { dB[k]++; });
                else
                    for_pairs_part<Cell>(MB[q], x, A < r, N, [&](int k) { dA[k]++; });
                for_pairs_cross<Cell>(x, MA[q], N, [&](int k) { cN[k]++; });
            }
        }
        int32_t *sp = slab.data();  // AA, AB, BB, AN, BN, NN
        count_pair(A, A, sp);
        if (A != B) {
            count_pair(A, B, sp + Kk);
            count_pair(B, B, sp + 2 * Kk);
            count_pair(B, Nw, sp + 4 * Kk);
        }
        count_pair(A, Nw, sp + 3 * Kk);
        count_pair(Nw, Nw, sp + 5 * Kk);
        scount = (int64_t)slab.size();
        return slab.data();
    }
    void apply() override {
        const int Kk = K;
        const int32_t *sp = slab.data();
        put_block(A, A, sp);
        if (A != B) {
            put_block(A, B, sp + Kk);
            put_block(B, B, sp + 2 * Kk);
            put_block(B, Nw, sp + 4 * Kk);
        }
        put_block(A, Nw, sp + 3 * Kk);
        put_block(Nw, Nw, sp + 5 * Kk);
        for (size_t u = 0; u < uni.size(); ++u) {
            uint32_t r = uni[u];
            st.partners++;
            const int32_t *dA = slab.data() + (6 + 3 * u) * Kk, *dB = dA + Kk, *cN = dB + Kk;
            auto sub = [&](uint32_t m, const int32_t *d) {
                auto it = table.find(Base::pkey(std::min(m, r), std::max(m, r)));
                if (it == table.end()) return;
                Block &b = it->second;
                for (int k = 0; k < Kk; ++k) b.cnt[k] = (uint16_t)(b.cnt[k] - d[k]);
                if (this->alive(b))
                    this->refresh_best(b);
                else
                    table.erase(it);
            };
            sub(A, dA);
            if (A != B) sub(B, dB);
            bool any = false;
            for (int k = 0; k < Kk; ++k) any |= cN[k] >= 2;
            if (any) put_block(r, Nw, cN);
        }
        st.table_peak = std::max(st.table_peak, (int64_t)table.size());
    }
    void finish(ChainOut &out) override {
        out = head;
        out.error = this->err ? E_FLOAT_DOMAIN : E_OK;
        for (auto &r : rows) out.row_lat.push_back(r.lat);
        out.col_start.assign(n_loc + 1, 0);
        for (int j = 0; j < n_loc; ++j) {
            for (size_t r = 0; r < cells.size(); ++r)
                if (cells[r][j]) {
                    out.dig_row.push_back((uint32_t)r);
                    out.dig_cell.push_back((uint64_t)O::plus(cells[r][j]) | ((uint64_t)O::minus(cells[r][j]) << 32));
                }
            out.col_start[j + 1] = (uint32_t)out.dig_row.size();
        }
        out.stats = st;
    }
};

std::unique_ptr<ShardEngine> make_model_shard(const ChainJob &job, int c0, int c1, double, void *) {
    std::vector<float> a(job.kernel, job.kernel + (size_t)job.n_in * job.n_out);
    std::vector<int8_t> s0, s1;
    center_matrix(a, job.n_in, job.n_out, s0, s1);
    uint32_t mx = 0;
    for (float v : a) mx = std::max(mx, (uint32_t)std::abs((int32_t)v));
    if (csd_width(mx) <= 16) return std::unique_ptr<ShardEngine>(new ShardChain<uint32_t>(job, c0, c1));
    return std::unique_ptr<ShardEngine>(new ShardChain<uint64_t>(job, c0, c1));
}

long long g_chains_run = 0;  // chains handed to the backend since the last reset (memoisation test)

class ModelBackend : public Backend {
  public:
    void run_chains(const ChainJob *jobs, ChainOut *outs, int n) override {
        g_chains_run += n;
        for (int i = 0; i < n; ++i) {
            // pick the cell width from the digit width, as the device does
            std::vector<float> a(jobs[i].kernel, jobs[i].kernel + (size_t)jobs[i].n_in * jobs[i].n_out);
            std::vector<int8_t> s0, s1;
            center_matrix(a, jobs[i].n_in, jobs[i].n_out, s0, s1);
            uint32_t mx = 0;
            for (float v : a) mx = std::max(mx, (uint32_t)std::abs((int32_t)v));
            if (csd_width(mx) <= 16) {
                Chain<uint32_t> c;
                c.run(jobs[i], outs[i]);
            } else {
                Chain<uint64_t> c;
                c.run(jobs[i], outs[i]);
            }
        }
    }
    void column_distances(const int32_t *aug, int n_in, int W, int64_t *d0, int64_t *d1) override {
        std::fill(d0, d0 + (size_t)W * W, 0);
        std::fill(d1, d1 + (size_t)W * W, 0);
        for (int i = 0; i < n_in; ++i)
            for (int a = 0; a < W; ++a)
                for (int b = 0; b < W; ++b) {
                    d0[(size_t)a * W + b] += naf_weight(aug[(size_t)i * W + a] - aug[(size_t)i * W + b]);
                    d1[(size_t)a * W + b] += naf_weight(aug[(size_t)i * W + a] + aug[(size_t)i * W + b]);
                }
    }
    int csd_decompose(const float *kernel, int n_in, int n_out, bool center, std::vector<int8_t> &csd,
                      std::vector<int8_t> &s0, std::vector<int8_t> &s1) override {
        std::vector<float> a(kernel, kernel + (size_t)n_in * n_out);
        if (center)
            center_matrix(a, n_in, n_out, s0, s1);
        else {
            s0.assign(n_in, 0);
            s1.assign(n_out, 0);
        }
        std::vector<int32_t> xi(a.size());
        for (size_t k = 0; k < a.size(); ++k) xi[k] = (int32_t)a[k];
        return int_to_csd(xi.data(), (int64_t)xi.size(), csd);
    }
    int int_to_csd(const int32_t *x, int64_t n, std::vector<int8_t> &csd) override {
        uint32_t mx = 0;
        for (int64_t i = 0; i < n; ++i) mx = std::max(mx, (uint32_t)std::abs(x[i]));
        int N = csd_width(mx);
        csd.assign((size_t)n * N, 0);
        for (int64_t i = 0; i < n; ++i) {
            uint32_t p, m;
            naf_masks(x[i], p, m);
            for (int b = 0; b < N; ++b) csd[(size_t)i * N + b] = (int8_t)(((p >> b) & 1) - ((m >> b) & 1));
        }
        return N;
    }
};

struct Result {
    PipeResult pipe;
    ChainStats stats;
};
thread_local std::string g_err;

}  // namespace

extern "C" {
const char *mdl_last_error() { return g_err.c_str(); }
int mdl_get_lsb_loc(float x) { return da::lsb_loc(x); }
int mdl_iceil_log2(float x) { return da::iceil_log2(x); }
void mdl_cost_add(const float *q0, const float *q1, int64_t shift, int sub, int adder_size, int carry_size, float *out2) {
    da::cost_add(da::QInt{q0[0], q0[1], q0[2]}, da::QInt{q1[0], q1[1], q1[2]}, shift, sub != 0, adder_size, carry_size, out2[0], out2[1]);
}
int mdl_int_arr_to_csd(const int32_t *x, int64_t n, int8_t *out) {
    ModelBackend be;
    std::vector<int8_t> csd;
    int N = be.int_to_csd(x, n, csd);
    if (out) std::memcpy(out, csd.data(), csd.size());
    return N;
}
int mdl_csd_decompose(const float *kernel, int64_t n_in, int64_t n_out, int do_center, int8_t *csd, int8_t *s0, int8_t *s1) {
    ModelBackend be;
    std::vector<int8_t> c, a, b;
    int N = be.csd_decompose(kernel, (int)n_in, (int)n_out, do_center != 0, c, a, b);
    if (csd) std::memcpy(csd, c.data(), c.size());
    if (s0) std::memcpy(s0, a.data(), a.size());
    if (s1) std::memcpy(s1, b.data(), b.size());
    return N;
}
void mdl_kernel_decompose(const float *kernel, int64_t n_in, int64_t n_out, int dc, float *m0, float *m1) {
    ModelBackend be;
    std::vector<float> a, b;
    da::kernel_decompose(be, kernel, (int)n_in, (int)n_out, dc, a, b);
    std::memcpy(m0, a.data(), a.size() * 4);
    std::memcpy(m1, b.data(), b.size() * 4);
}
void *mdl_solve(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                int decompose_dc, const float *qints3, const float *lats, int adder_size, int carry_size, int search_all) {
    try {
        ModelBackend be;
        da::Problem p;
        p.kernel = kernel;
        p.n_in = (int)n_in;
        p.n_out = (int)n_out;
        p.opt.method0 = method0;
        p.opt.method1 = method1;
        p.opt.hard_dc = hard_dc;
        p.opt.decompose_dc = decompose_dc;
        if (qints3)
            for (int64_t i = 0; i < n_in; ++i) p.opt.qints.push_back(da::QInt{qints3[3 * i], qints3[3 * i + 1], qints3[3 * i + 2]});
        if (lats) p.opt.lats.assign(lats, lats + n_in);
        p.opt.adder_size = adder_size;
        p.opt.carry_size = carry_size;
        p.opt.search_all = search_all != 0;
        std::vector<da::ChainStats> st;
        auto res = da::solve_batch(be, {p}, &st);
        auto *r = new Result{std::move(res[0]), st.empty() ? da::ChainStats{} : st[0]};
        return r;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
// batch of problems sharing the option set (the shape of da_solve_batch); qints3 / lats: per-problem pointers or NULL
int mdl_solve_batch(int count, const float *const *kernels, const int64_t *n_in, const int64_t *n_out, const char *method0, const char *method1,
                    int hard_dc, int decompose_dc, const float *const *qints3, const float *const *lats, int adder_size, int carry_size,
                    int search_all, void **results) {
    try {
        ModelBackend be;
        std::vector<da::Problem> probs((size_t)count);
        for (int i = 0; i < count; ++i) {
            da::Problem &p = probs[i];
            p.kernel = kernels[i];
            p.n_in = (int)n_in[i];
            p.n_out = (int)n_out[i];
            p.opt.method0 = method0;
            p.opt.method1 = method1;
            p.opt.hard_dc = hard_dc;
            p.opt.decompose_dc = decompose_dc;
            if (qints3 && qints3[i])
                for (int64_t r = 0; r < n_in[i]; ++r) p.opt.qints.push_back(da::QInt{qints3[i][3 * r], qints3[i][3 * r + 1], qints3[i][3 * r + 2]});
            if (lats && lats[i]) p.opt.lats.assign(lats[i], lats[i] + n_in[i]);
            p.opt.adder_size = adder_size;
            p.opt.carry_size = carry_size;
            p.opt.search_all = search_all != 0;
        }
        std::vector<da::ChainStats> st;
        auto res = da::solve_batch(be, probs, &st);
        for (int i = 0; i < count; ++i) results[i] = new Result{std::move(res[i]), i < (int)st.size() ? st[i] : da::ChainStats{}};
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}
// One solve with every greedy chain column-sharded over the ranks of the caller's process group (the shape of
// da_solve_sharded): `allreduce` is called for every exchange; stats3 = {sharded chains, greedy steps, all-reduce calls}.
static std::atomic<int> g_mdl_comm_aborted{0};
void mdl_comm_abort(void) { g_mdl_comm_aborted.store(1); }
void *mdl_solve_sharded(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                        int decompose_dc, const float *qints3, const float *lats, int adder_size, int carry_size, int search_all,
                        int rank, int world, da::allreduce_i32_fn allreduce, void *ctx, int64_t *stats3) {
    try {
        ModelBackend inner;
        da::ShardComm comm;
        comm.rank = rank;
        comm.world = world;
        comm.allreduce = allreduce;
        comm.ctx = ctx;
        g_mdl_comm_aborted.store(0);
        comm.aborted = &g_mdl_comm_aborted;
        da::ShardedBackend be(inner, comm, make_model_shard, nullptr);
        da::Problem p;
        p.kernel = kernel;
        p.n_in = (int)n_in;
        p.n_out = (int)n_out;
        p.opt.method0 = method0;
        p.opt.method1 = method1;
        p.opt.hard_dc = hard_dc;
        p.opt.decompose_dc = decompose_dc;
        if (qints3)
            for (int64_t i = 0; i < n_in; ++i) p.opt.qints.push_back(da::QInt{qints3[3 * i], qints3[3 * i + 1], qints3[3 * i + 2]});
        if (lats) p.opt.lats.assign(lats, lats + n_in);
        p.opt.adder_size = adder_size;
        p.opt.carry_size = carry_size;
        p.opt.search_all = search_all != 0;
        std::vector<da::ChainStats> st;
        auto res = da::solve_batch(be, {p}, &st);
        if (stats3) {
            stats3[0] = be.sharded_chains;
            stats3[1] = be.sharded_steps;
            stats3[2] = be.comm().calls;
        }
        return new Result{std::move(res[0]), st.empty() ? da::ChainStats{} : st[0]};
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
long long mdl_chains_run(int reset) {
    long long v = g_chains_run;
    if (reset) g_chains_run = 0;
    return v;
}
int mdl_n_stages(void *h) { return (int)((Result *)h)->pipe.stages.size(); }
int mdl_picked(void *h) { return ((Result *)h)->pipe.picked; }
void mdl_stage_info(void *h, int s, int64_t *info) {
    const da::StageResult &st = ((Result *)h)->pipe.stages[s];
    info[0] = st.n_in;
    info[1] = st.n_out;
    info[2] = (int64_t)st.ops.size();
    info[3] = st.carry_size;
    info[4] = st.adder_size;
}
void mdl_stage_copy(void *h, int s, int64_t *inp_shifts, int64_t *out_idxs, int64_t *out_shifts, int64_t *out_negs,
                    int64_t *ops_i, float *ops_f) {
    const da::StageResult &st = ((Result *)h)->pipe.stages[s];
    std::memcpy(inp_shifts, st.inp_shifts.data(), st.inp_shifts.size() * 8);
    std::memcpy(out_idxs, st.out_idxs.data(), st.out_idxs.size() * 8);
    std::memcpy(out_shifts, st.out_shifts.data(), st.out_shifts.size() * 8);
    std::memcpy(out_negs, st.out_negs.data(), st.out_negs.size() * 8);
    for (size_t k = 0; k < st.ops.size(); ++k) {
        const da::OpRec &o = st.ops[k];
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
// stats[0..7]: iterations, digits0, blocks0, rebuilds, table_peak, scan_slots, partners, matches
void mdl_stats(void *h, int64_t *s) {
    const da::ChainStats &t = ((Result *)h)->stats;
    int64_t v[8] = {t.iterations, t.digits0, t.blocks0, t.rebuilds, t.table_peak, t.scan_slots, t.partners, t.matches};
    std::memcpy(s, v, sizeof v);
}
void mdl_free(void *h) { delete (Result *)h; }
}
