// cmvm_host.cc -- host side of the MI355X CMVM solver (see cmvm_host.h for the device/host split).
// Reference behaviour restated here: api.cc (option resolution, candidate search), mat_decompose.cc
// (MST + m0/m1 assembly), cmvm_core.cc:75-225 (adder trees), state_opr.cc:8-67 (interval / cost model).

#include "cmvm_host.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <unordered_map>
#include <thread>

namespace da {

int parse_method(const std::string &name) {
    static const char *names[] = {"mc", "mc-dc", "mc-pdc", "wmc", "wmc-dc", "wmc-pdc", "dummy"};
    for (int i = 0; i < 7; ++i)
        if (name == names[i]) return i;
    return -1;
}

// ------------------------------------------------------------------------------- scalar helpers
QInt qint_add(const QInt &a, const QInt &b, int64_t shift, bool neg_a, bool neg_b) {
    float lo0 = neg_a ? -a.hi : a.lo, hi0 = neg_a ? -a.lo : a.hi;
    float lo1 = neg_b ? -b.hi : b.lo, hi1 = neg_b ? -b.lo : b.hi;
    float scale = (float)std::pow(2.0, (double)shift);
    lo1 *= scale;
    hi1 *= scale;
    float st1 = b.step * scale;
    return QInt{lo0 + lo1, hi0 + hi1, std::min(a.step, st1)};
}

void cost_add(const QInt &a, const QInt &b, int64_t shift, bool sub, int adder_size, int carry_size, float &dlat,
              float &cost) {
    if (adder_size < 0 && carry_size < 0) {
        dlat = cost = 1.0f;
        return;
    }
    int adder = adder_size < 0 ? 65535 : adder_size, carry = carry_size < 0 ? 65535 : carry_size;
    float scale = (float)std::pow(2.0, (double)shift);
    float lo1 = (sub ? b.hi : b.lo) * scale, hi1 = (sub ? b.lo : b.hi) * scale, st1 = b.step * scale;
    float hi0 = a.hi + a.step;
    hi1 += st1;
    float frac_bits = -std::log2(std::max(a.step, st1));
    float int_bits = std::ceil(std::log2(std::max({std::abs(a.lo), std::abs(lo1), std::abs(hi0), std::abs(hi1)})));
    int sign_bit = (a.lo < 0 || b.lo < 0) ? 1 : 0;
    float width = sign_bit + int_bits + frac_bits;
    dlat = std::ceil(width / carry);
    cost = std::ceil(width / adder);
}

Log2Table measure_log2_table() {
    Log2Table t;
    std::memset(t.tie, 0, sizeof t.tie);
    for (int e = -126; e <= 127; ++e) {
        int run = 0;
        for (int k = 1; k < 255; ++k) {
            float x = std::ldexp(1.0f + (float)k * 0x1p-23f, e);
            if (std::log2(x) == (float)e)
                run = k;
            else
                break;
        }
        t.tie[e + 150] = (uint8_t)run;
    }
    return t;
}

void StepLog2Host::build(const QInt *q, int n_in) {
    mant.clear();
    tab.clear();
    if (!q) return;
    for (int i = 0; i < n_in; ++i) {
        if (q[i].lo == 0.0f && q[i].hi == 0.0f) continue;
        const uint32_t b = f2u(q[i].step), m = b & 0x7FFFFFu, e = (b >> 23) & 0xFFu;
        if ((b >> 31) || m == 0 || e == 0 || e == 255) continue;
        if (std::find(mant.begin(), mant.end(), m) != mant.end()) continue;
        mant.push_back(m);
    }
    tab.assign(mant.size() * 256, 0.0f);
    for (size_t i = 0; i < mant.size(); ++i)
        for (uint32_t e = 1; e < 255; ++e) tab[i * 256 + e] = -std::log2(u2f((e << 23) | mant[i]));  // float overload, as state_opr.cc:57
}

void center_matrix(std::vector<float> &a, int n_in, int n_out, std::vector<int8_t> &s0, std::vector<int8_t> &s1) {
    s0.assign(n_in, 0);
    s1.assign(n_out, 0);
    for (int j = 0; j < n_out; ++j) {
        int low = 127;
        for (int i = 0; i < n_in; ++i) low = std::min(low, lsb_loc(a[(size_t)i * n_out + j]));
        s1[j] = (int8_t)low;
        double scale = std::pow(2.0, -low);
        for (int i = 0; i < n_in; ++i) a[(size_t)i * n_out + j] = (float)(a[(size_t)i * n_out + j] * scale);
    }
    for (int i = 0; i < n_in; ++i) {
        int low = 127;
        for (int j = 0; j < n_out; ++j) low = std::min(low, lsb_loc(a[(size_t)i * n_out + j]));
        s0[i] = (int8_t)low;
        double scale = std::pow(2.0, -low);
        for (int j = 0; j < n_out; ++j) a[(size_t)i * n_out + j] = (float)(a[(size_t)i * n_out + j] * scale);
    }
}

// ------------------------------------------------------------------------------- stage 1
namespace {

struct Stage1 {  // everything about a matrix that does not depend on decompose_dc
    int n_in = 0, n_out = 0, W = 0;
    std::vector<float> centered, aug;
    std::vector<int8_t> s0, s1;
    std::vector<int64_t> dist, sign;
    bool have_dist = false;
};

void stage1_prepare(Stage1 &s, const float *kernel, int n_in, int n_out) {
    s.n_in = n_in;
    s.n_out = n_out;
    s.W = n_out + 1;
    s.centered.assign(kernel, kernel + (size_t)n_in * n_out);
    center_matrix(s.centered, n_in, n_out, s.s0, s.s1);
}

// column distances of the matrix augmented with a zero column (only decompositions with dc != -1 need them).  Called from one
// thread per Stage1 object: before the candidates of a problem are split on several threads, or from kernel_decompose().
void stage1_distances(Backend &be, Stage1 &s) {
    if (s.have_dist) return;
    size_t W = (size_t)s.W;
    s.aug.assign((size_t)s.n_in * W, 0.0f);
    for (int i = 0; i < s.n_in; ++i) std::copy_n(&s.centered[(size_t)i * s.n_out], s.n_out, &s.aug[(size_t)i * W + 1]);
    std::vector<int32_t> ai(s.aug.size());
    for (size_t k = 0; k < ai.size(); ++k) ai[k] = (int32_t)s.aug[k];
    std::vector<int64_t> d0(W * W), d1(W * W);
    be.column_distances(ai.data(), s.n_in, s.W, d0.data(), d1.data());
    s.dist.resize(W * W);
    s.sign.resize(W * W);
    for (size_t k = 0; k < W * W; ++k) {
        s.sign[k] = d1[k] - d0[k] < 0 ? -1 : 1;
        s.dist[k] = std::min(d0[k], d1[k]);
    }
    s.have_dist = true;
}

// Prim's tree from vertex 0 with the reference's scan order and optional depth cap (mat_decompose.cc:6-60)
std::vector<std::pair<int, int>> spanning_tree(const std::vector<int64_t> &cost, int V, int dc) {
    auto edge_lat = [&](int i, int j) { return std::ceil(std::log2((float)std::max<int64_t>(cost[(size_t)i * V + j], 1))); };
    std::vector<char> in_tree(V, 0);
    std::vector<int32_t> depth(V, 0);
    in_tree[0] = 1;
    float cap = -1.0f;
    if (dc >= 0) {
        float top = (float)*std::max_element(cost.begin(), cost.begin() + V);
        cap = (float)((std::pow(2.0, dc) - 1) + std::ceil(std::log2(top + 1e-32)));
    }
    const int64_t blocked = std::numeric_limits<int64_t>::max() / 2;
    std::vector<std::pair<int, int>> edges;
    edges.reserve(V > 0 ? V - 1 : 0);
    for (int step = 1; step < V; ++step) {
        int64_t best = std::numeric_limits<int64_t>::max();
        int bi = -1, bj = -1;
        for (int i = 0; i < V; ++i) {
            if (in_tree[i]) continue;
            for (int j = 0; j < V; ++j) {
                if (!in_tree[j]) continue;
                int64_t c = cost[(size_t)i * V + j];
                if (dc >= 0 && std::max(edge_lat(i, j), (float)depth[j]) + 1 > cap) c = blocked;
                if (c < best) {
                    best = c;
                    bi = i;
                    bj = j;
                }
            }
        }
        in_tree[bi] = 1;
        depth[bi] = (int32_t)(std::max(edge_lat(bi, bj), (float)depth[bj]) + 1);
        edges.emplace_back(bj, bi);
    }
    return edges;
}

void stage1_split(Backend &be, Stage1 &s, int dc, std::vector<float> &m0, std::vector<float> &m1) {
    int n_in = s.n_in, n_out = s.n_out, W = s.W;
    m0.assign((size_t)n_in * n_out, 0.0f);
    m1.assign((size_t)n_out * n_out, 0.0f);
    if (dc == -1) {  // no decomposition: m0 = centred kernel, m1 = identity (mat_decompose.cc:101-107)
        m0 = s.centered;
        for (int j = 0; j < n_out; ++j) m1[(size_t)j * n_out + j] = 1.0f;
    } else {
        stage1_distances(be, s);
        auto edges = spanning_tree(s.dist, W, dc);
        int used = 0;
        std::vector<float> delta(n_in), combo(n_out);
        for (auto [from, to] : edges) {
            float sg = (float)s.sign[(size_t)to * W + from];
            bool nonzero = false;
            for (int i = 0; i < n_in; ++i) {
                delta[i] = s.aug[(size_t)i * W + to] - s.aug[(size_t)i * W + from] * sg;
                nonzero |= delta[i] != 0.0f;
            }
            for (int r = 0; r < n_out; ++r) combo[r] = from != 0 ? m1[(size_t)r * n_out + (from - 1)] * sg : 0.0f;
            if (nonzero) {
                combo[used] = 1.0f;
                for (int i = 0; i < n_in; ++i) m0[(size_t)i * n_out + used] = delta[i];
                ++used;
            }
            for (int r = 0; r < n_out; ++r) m1[(size_t)r * n_out + (to - 1)] = combo[r];
        }
    }
    for (int i = 0; i < n_in; ++i) {
        float sc = std::pow(2.0f, (float)s.s0[i]);
        for (int j = 0; j < n_out; ++j) m0[(size_t)i * n_out + j] *= sc;
    }
    for (int j = 0; j < n_out; ++j) {
        float sc = std::pow(2.0f, (float)s.s1[j]);
        for (int r = 0; r < n_out; ++r) m1[(size_t)r * n_out + j] *= sc;
    }
}

}  // namespace

void kernel_decompose(Backend &be, const float *kernel, int n_in, int n_out, int dc, std::vector<float> &m0,
                      std::vector<float> &m1) {
    Stage1 s;
    stage1_prepare(s, kernel, n_in, n_out);
    stage1_split(be, s, dc, m0, m1);
}

// ------------------------------------------------------------------------------- adder trees
namespace {

struct Term {
    float lat;
    int64_t neg, align;
    QInt q;
    int64_t id, shift;
};
inline bool term_after(const Term &x, const Term &y) {  // x > y in (lat, neg, align, q.lo, q.hi, q.step, id, shift)
    if (x.lat != y.lat) return x.lat > y.lat;
    if (x.neg != y.neg) return x.neg > y.neg;
    if (x.align != y.align) return x.align > y.align;
    if (x.q.lo != y.q.lo) return x.q.lo > y.q.lo;
    if (x.q.hi != y.q.hi) return x.q.hi > y.q.hi;
    if (x.q.step != y.q.step) return x.q.step > y.q.step;
    if (x.id != y.id) return x.id > y.id;
    return x.shift > y.shift;
}
inline int64_t magnitude_bits(const QInt &q) { return (int64_t)std::log2(std::max(std::abs(q.hi + q.step), std::abs(q.lo))); }

}  // namespace

// Chains without a greedy loop never need the device: the "dummy" method (the minimal-latency probe, api.cc:11-26: adder
// trees straight from the CSD digits) and matrices in which no column holds two digits, so that no digit pair exists at
// all -- above all the identity second stage of every decompose_dc = -1 solve.  The result is what the device kernels
// produce for such a chain (k_prepare + k_init_cells + k_extract): centring shifts, digit width, the digits of every
// column in row order.  Returns false when the chain has to run on the device.
static bool host_chain(const ChainJob &job, ChainOut &out) {
    const int n_in = job.n_in, n_out = job.n_out;
    if (job.method != M_DUMMY && job.method >= 0) {
        // quick way out for the ordinary chain: a column with two non-zero entries in live rows holds at least two digits, so
        // the chain has a greedy loop and belongs to the device (found within the first rows of a dense matrix; the full
        // examination below copies and centres the matrix)
        std::vector<uint8_t> seen((size_t)n_out, 0);
        for (int i = 0; i < n_in; ++i) {
            if (job.qints[i].lo == 0.0f && job.qints[i].hi == 0.0f) continue;
            const float *row = job.kernel + (size_t)i * n_out;
            for (int j = 0; j < n_out; ++j)
                if (row[j] != 0.0f) {
                    if (seen[j]) return false;
                    seen[j] = 1;
                }
        }
    }
    std::vector<float> a(job.kernel, job.kernel + (size_t)n_in * n_out);
    std::vector<int8_t> s0, s1;
    center_matrix(a, n_in, n_out, s0, s1);
    std::vector<uint8_t> dead(n_in);
    for (int i = 0; i < n_in; ++i) dead[i] = job.qints[i].lo == 0.0f && job.qints[i].hi == 0.0f;
    uint32_t mx = 0;
    long long digits0 = 0;
    std::vector<int> dcol(n_out, 0);
    for (int i = 0; i < n_in; ++i)
        for (int j = 0; j < n_out; ++j) {
            const int32_t x = (int32_t)a[(size_t)i * n_out + j];
            mx = std::max(mx, (uint32_t)(x < 0 ? -(int64_t)x : x));
            if (!dead[i] && x != 0) dcol[j] += naf_weight(x);
        }
    if (job.method != M_DUMMY) {
        if (job.method < 0) return false;  // unknown method string: the device path decides whether it has to raise
        for (int j = 0; j < n_out; ++j)
            if (dcol[j] > 1) return false;
    }
    const int n_bits = csd_width(mx);
    if (n_bits > 30) return false;  // rejected with a message by the device path
    out = ChainOut{};
    out.n_bits = n_bits;
    out.shift0 = s0;
    out.shift1 = s1;
    out.row_lat.assign(job.lats, job.lats + n_in);
    out.col_start.assign((size_t)n_out + 1, 0);
    for (int j = 0; j < n_out; ++j) {
        for (int i = 0; i < n_in; ++i) {
            const int32_t x = (int32_t)a[(size_t)i * n_out + j];
            if (dead[i] || x == 0) continue;
            uint32_t p, m;
            naf_masks(x, p, m);
            out.dig_row.push_back((uint32_t)i);
            out.dig_cell.push_back((uint64_t)p | ((uint64_t)m << 32));
        }
        out.col_start[j + 1] = (uint32_t)out.dig_row.size();
        digits0 += dcol[j];
    }
    out.stats.digits0 = digits0;
    return true;
}

namespace {
struct OpListStash {
    std::mutex mu;
    std::vector<std::vector<OpRec>> lists;
    size_t bytes = 0;
    static constexpr size_t MAX_BYTES = (size_t)1 << 30, MAX_LISTS = 1024, MIN_KEEP = 4096;  // ops; smaller lists are not worth keeping
};
OpListStash &op_list_stash() {
    // leaked: results may be released during interpreter shutdown.  fork(): the mutex is taken before the fork and released on
    // both sides, so that a child never inherits it locked by a thread that does not exist there
    static OpListStash *s = [] {
        pthread_atfork([] { op_list_stash().mu.lock(); }, [] { op_list_stash().mu.unlock(); }, [] { op_list_stash().mu.unlock(); });
        return new OpListStash();
    }();
    return *s;
}
}  // namespace

std::vector<OpRec> take_op_list(size_t capacity) {
    OpListStash &st = op_list_stash();
    {
        std::lock_guard<std::mutex> lk(st.mu);
        int best = -1;
        for (int k = (int)st.lists.size() - 1; k >= 0; --k)  // the smallest one that is large enough
            if (st.lists[k].capacity() >= capacity && (best < 0 || st.lists[k].capacity() < st.lists[best].capacity())) best = k;
        if (best >= 0) {
            std::vector<OpRec> v = std::move(st.lists[best]);
            st.lists[best] = std::move(st.lists.back());
            st.lists.pop_back();
            st.bytes -= v.capacity() * sizeof(OpRec);
            v.clear();
            return v;
        }
    }
    std::vector<OpRec> v;
    v.reserve(capacity);
    return v;
}

void recycle_op_list(std::vector<OpRec> &&ops) {
    if (ops.capacity() < OpListStash::MIN_KEEP) return;
    OpListStash &st = op_list_stash();
    std::vector<OpRec> drop;  // released outside the lock
    std::lock_guard<std::mutex> lk(st.mu);
    const size_t b = ops.capacity() * sizeof(OpRec);
    if (st.bytes + b > OpListStash::MAX_BYTES || st.lists.size() >= OpListStash::MAX_LISTS) {
        drop = std::move(ops);
        return;
    }
    ops.clear();
    st.bytes += b;
    st.lists.push_back(std::move(ops));
}

// Part 1 (one thread per chain): input and greedy-pick op records, sizes and ids of the per-column adder trees.
// Part 2 (any thread, any order, disjoint column ranges): the trees, written straight into their final places.
StageResult finalize_prepare(const ChainJob &job, const ChainOut &out, std::vector<int64_t> &first_op) {
    if (out.error == E_REMOTE_ERROR) throw std::runtime_error("column-sharded CMVM chain failed on another rank (not a capacity error), or the ranks fell out of step");
    if (out.error != E_OK) throw std::runtime_error("CMVM chain failed on the device (error " + std::to_string(out.error) + ")");
    StageResult r;
    r.n_in = job.n_in;
    r.n_out = job.n_out;
    r.adder_size = job.adder_size;
    r.carry_size = job.carry_size;
    r.inp_shifts.assign(out.shift0.begin(), out.shift0.end());
    size_t n_iter = out.picks.size() / 4;
    // a column with t terms (digits) emits exactly t - 1 tree ops: the size of the op list is known before anything is built
    const int n_out = job.n_out;
    first_op.assign((size_t)n_out + 1, 0);  // tree ops of the columns before j
    for (int j = 0; j < n_out; ++j) {
        int64_t terms = 0;
        for (uint32_t k = out.col_start[j]; k < out.col_start[j + 1]; ++k)
            terms += __builtin_popcountll(((uint64_t)(uint32_t)out.dig_cell[k]) | (out.dig_cell[k] >> 32));
        first_op[j + 1] = first_op[j] + (terms > 1 ? terms - 1 : 0);
    }
    r.ops = take_op_list((size_t)job.n_in + n_iter + (size_t)first_op[n_out]);
    for (int i = 0; i < job.n_in; ++i) r.ops.push_back(OpRec{i, -1, -1, 0, job.qints[i], job.lats[i], 0.0f});
    // op records of the greedy picks, with the host libm (state_opr.cc:211-225)
    for (size_t t = 0; t < n_iter; ++t) {
        int64_t a = out.picks[4 * t], b = out.picks[4 * t + 1];
        bool sub = out.picks[4 * t + 2] != 0;
        int64_t shift = out.picks[4 * t + 3];
        const OpRec &oa = r.ops[a], &ob = r.ops[b];
        float dlat, cost;
        cost_add(oa.q, ob.q, shift, sub, job.adder_size, job.carry_size, dlat, cost);
        float lat = std::max(oa.latency, ob.latency) + dlat;
        float seen = out.row_lat[job.n_in + t];
        if (std::memcmp(&lat, &seen, 4) != 0 && !(lat != lat && seen != seen))
            throw std::runtime_error("device latency model diverged from the host libm at iteration " + std::to_string(t));
        r.ops.push_back(OpRec{a, b, (int64_t)sub, shift, qint_add(oa.q, ob.q, shift, false, sub), lat, cost});
    }
    // One min-heap reduction per output column follows (finalize_columns, cmvm_core.cc:103-210).  The columns are independent:
    // a tree op refers to pick / input ops (final already) and to earlier ops of its own column only, and the ids of every
    // column are known before any tree is built (prefix sum of the term counts above; the reference numbers the ops column
    // after column, cmvm_core.cc:101,199-203): the op list is sized once here and column ranges are reduced on any thread
    // straight into their final places -- no per-thread lists, no renumbering, no concatenation.
    const int64_t n_fixed = (int64_t)r.ops.size();
    r.ops.resize((size_t)(n_fixed + first_op[n_out]));
    r.out_idxs.assign((size_t)n_out, -1);
    r.out_shifts.assign((size_t)n_out, 0);
    r.out_negs.assign((size_t)n_out, 0);
    return r;
}

void finalize_columns(const ChainJob &job, const ChainOut &out, StageResult &r, const std::vector<int64_t> &first_op, int j0, int j1) {
    const int64_t n_fixed = (int64_t)r.ops.size() - first_op[job.n_out];
    {
        std::vector<Term> heap;
        auto cmp = [](const Term &x, const Term &y) { return term_after(x, y); };
        for (int j = j0; j < j1; ++j) {
            heap.clear();
            for (uint32_t k = out.col_start[j]; k < out.col_start[j + 1]; ++k) {
                uint64_t cell = out.dig_cell[k];
                uint32_t plus = (uint32_t)cell, minus = (uint32_t)(cell >> 32), any = plus | minus;
                int64_t row = out.dig_row[k];
                while (any) {
                    int pos = __builtin_ctz(any);
                    any &= any - 1;
                    const OpRec &o = r.ops[row];
                    heap.push_back(Term{o.latency, (int64_t)((minus >> pos) & 1), magnitude_bits(o.q) + pos, o.q, row, pos});
                }
            }
            if (heap.empty()) {
                r.out_shifts[j] = out.shift1[j];  // out_idxs -1, not negated
                continue;
            }
            if (heap.size() == 1) {
                r.out_idxs[j] = heap[0].id;
                r.out_shifts[j] = (int64_t)out.shift1[j] + heap[0].shift;
                r.out_negs[j] = heap[0].neg;
                continue;
            }
            int64_t id = n_fixed + first_op[j];  // this column's ops: [id, n_fixed + first_op[j + 1])
            std::make_heap(heap.begin(), heap.end(), cmp);
            while (heap.size() > 1) {
                std::pop_heap(heap.begin(), heap.end(), cmp);
                Term first = heap.back();
                heap.pop_back();
                std::pop_heap(heap.begin(), heap.end(), cmp);
                Term second = heap.back();
                heap.pop_back();
                // the result is anchored on the non-negated operand when the first one is negative
                const Term &base = first.neg ? second : first, &other = first.neg ? first : second;
                int64_t sh = other.shift - base.shift;
                bool sub_op = first.neg ? (second.neg == 0) : (second.neg != 0);
                QInt q = qint_add(base.q, other.q, sh, base.neg != 0, other.neg != 0);
                float dlat, cost;
                cost_add(base.q, other.q, sh, sub_op, job.adder_size, job.carry_size, dlat, cost);
                float lat = std::max(first.lat, second.lat) + dlat;
                r.ops[id] = OpRec{base.id, other.id, (int64_t)sub_op, sh, q, lat, cost};
                heap.push_back(Term{lat, first.neg & second.neg, magnitude_bits(q) + base.shift, q, id, base.shift});
                std::push_heap(heap.begin(), heap.end(), cmp);
                ++id;
            }
            if (id != n_fixed + first_op[j + 1]) throw std::runtime_error("adder tree of a column emitted an unexpected number of ops (internal error)");
            r.out_idxs[j] = id - 1;
            r.out_negs[j] = heap[0].neg;
            r.out_shifts[j] = (int64_t)out.shift1[j] + heap[0].shift;
        }
    }
}

StageResult finalize_chain(const ChainJob &job, const ChainOut &out, int inner_threads) {
    std::vector<int64_t> first_op;
    StageResult r = finalize_prepare(job, out, first_op);
    const int n_out = job.n_out;
    const int n_chunks = inner_threads > 1 ? std::min(n_out, inner_threads * 4) : 1;
    auto reduce_chunk = [&](int c) {
        finalize_columns(job, out, r, first_op, (int)((long long)n_out * c / n_chunks), (int)((long long)n_out * (c + 1) / n_chunks));
    };
    if (n_chunks <= 1)
        finalize_columns(job, out, r, first_op, 0, n_out);
    else {
        std::atomic<int> next{0};
        std::exception_ptr err;
        std::mutex err_mu;
        auto work = [&] {
            for (int c = next++; c < n_chunks; c = next++) {
                try {
                    reduce_chunk(c);
                } catch (...) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (!err) err = std::current_exception();
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < std::min(inner_threads, n_chunks); ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (err) std::rethrow_exception(err);
    }
    return r;
}

// ------------------------------------------------------------------------------- search orchestration
namespace {

// run fn(i) for i in [0, n) on the host threads (the per-chain host work -- centring, MST, adder trees -- is independent).
// The threads are created once and parked on a condition variable: a batch call goes through seven such loops and creating
// 64 threads costs 1-3 ms each time (measured), i.e. up to 2 % of a 64-chain step spent in pthread_create.  The object is
// leaked on purpose (no static destructor joins threads at exit) and abandoned in a forked child, which starts its own.
class HostPool {
  public:
    static HostPool &get() {
        // fork(): the creation mutex is held across the fork (never inherited locked); the child drops the parent's pool -- its
        // threads do not exist there -- and starts its own on first use
        static std::once_flag once;
        std::call_once(once, [] {
            pthread_atfork([] { make_mu().lock(); }, [] { make_mu().unlock(); },
                           [] {
                               instance().store(nullptr);
                               make_mu().unlock();
                           });
        });
        HostPool *p = instance().load();
        if (!p) {
            std::lock_guard<std::mutex> lk(make_mu());
            p = instance().load();
            if (!p) {
                p = new HostPool();
                instance().store(p);
            }
        }
        return *p;
    }
    // false: the pool is in use (a nested or concurrent loop) -- the caller runs the loop on threads of its own
    template <class Fn> bool run(size_t n, Fn &&fn) {
        if (busy_.test_and_set(std::memory_order_acquire)) return false;
        struct Release {
            std::atomic_flag &f;
            ~Release() { f.clear(std::memory_order_release); }
        } release{busy_};
        std::exception_ptr err;
        std::mutex err_mu;
        std::atomic<size_t> next{0};
        std::function<void()> body = [&] {
            for (size_t i = next++; i < n; i = next++) {
                try {
                    fn(i);
                } catch (...) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (!err) err = std::current_exception();
                }
            }
        };
        const int helpers = (int)std::min<size_t>((size_t)cap_, n - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            // threads are started when a loop first needs them: a single small solve starts a handful, not 255
            while ((int)threads_.size() < helpers) {
                const int t = (int)threads_.size();
                threads_.emplace_back([this, t] { worker(t); });
                threads_.back().detach();
            }
            body_ = &body;
            wanted_ = helpers;
            pending_ = helpers;
            ++epoch_;
        }
        cv_work_.notify_all();
        if (helpers > NARROW) cv_wide_.notify_all();  // the threads beyond the first 63 are woken for wide loops only
        body();  // the calling thread works too
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return pending_ == 0; });
            body_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
        return true;
    }
    int capacity() const { return cap_ + 1; }

  private:
    static std::atomic<HostPool *> &instance() {
        static std::atomic<HostPool *> p{nullptr};
        return p;
    }
    static std::mutex &make_mu() {
        static std::mutex *m = new std::mutex();
        return *m;
    }
    HostPool() {
        // The cores THIS process may run on (one process per GPU: multi_gpu.init() gives every rank its slice of the host's
        // cores, so eight ranks do not start eight pools of 256 threads), at most 256 -- the GPU box's host has 256 for the
        // tree pieces of a 64-chain batch.
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) hw = (unsigned)CPU_COUNT(&set);
        cap_ = (int)std::max(1u, std::min(hw ? hw : 1u, 256u)) - 1;
        if (const char *e = std::getenv("DA4ML_HOST_THREADS")) cap_ = std::max(1, std::min(1024, std::atoi(e))) - 1;  // cap / test hook
    }
    void worker(int index) {
        uint64_t seen = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            seen = epoch_ - 1;  // started inside run(), before the epoch of that loop was published: take part in it
        }
        for (;;) {
            std::function<void()> *body = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                (index < NARROW ? cv_work_ : cv_wide_).wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (index < wanted_) body = body_;
            }
            if (!body) continue;
            (*body)();
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
    std::mutex mu_;
    std::atomic_flag busy_ = ATOMIC_FLAG_INIT;
    static constexpr int NARROW = 63;
    std::condition_variable cv_work_, cv_wide_, cv_done_;
    std::vector<std::thread> threads_;
    std::function<void()> *body_ = nullptr;
    uint64_t epoch_ = 0;
    int cap_ = 0, wanted_ = 0, pending_ = 0;
};

template <class Fn> void parallel_for(size_t n, Fn &&fn) {
    if (n <= 1) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    if (HostPool::get().run(n, fn)) return;
    unsigned hw = std::thread::hardware_concurrency();
    size_t workers = std::min<size_t>(n, std::max(1u, std::min(hw ? hw : 1u, 64u)));
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex err_mu;
    std::vector<std::thread> pool;
    for (size_t w = 0; w < workers; ++w)
        pool.emplace_back([&] {
            for (size_t i = next++; i < n; i = next++) {
                try {
                    fn(i);
                } catch (...) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (!err) err = std::current_exception();
                }
            }
        });
    for (auto &t : pool) t.join();
    if (err) std::rethrow_exception(err);
}

bool has_dc_suffix(const std::string &m) { return m.size() >= 2 && m.compare(m.size() - 2, 2, "dc") == 0; }

struct Candidate {  // one _solve() of the reference (api.cc:28-145) as a resumable state machine
    int problem = 0;
    std::string method0, method1;
    int hard_dc = -1, decompose_dc = -2;
    float allowed = std::numeric_limits<float>::infinity();
    enum Phase { NEED_MINLAT, NEED_STAGE0, NEED_STAGE1, DONE } phase = NEED_STAGE0;
    std::vector<float> m0, m1;
    StageResult sol0, sol1;
    std::vector<QInt> q_mid;
    std::vector<float> lat_mid;
};

struct ProblemState {
    const Problem *p = nullptr;
    std::vector<QInt> qints;
    std::vector<float> lats;
    std::shared_ptr<Stage1> s1p;  // shared by the problems of a batch that have the same matrix (the tracer's row loop)
    bool minlat_known = false;
    float minlat = 0.0f;
    std::vector<int> cand;  // indices into the candidate array
};

float max_out_latency(const StageResult &s) {
    float top = 0.0f;
    for (auto idx : s.out_idxs) top = std::max(top, idx >= 0 ? s.ops[idx].latency : 0.0f);
    return top;
}

}  // namespace

namespace {
std::vector<PipeResult> solve_batch_unique(Backend &be, const std::vector<Problem> &problems, std::vector<ChainStats> *stats);

bool same_problem(const Problem &a, const Problem &b) {
    const SolveOptions &x = a.opt, &y = b.opt;
    if (a.n_in != b.n_in || a.n_out != b.n_out || x.hard_dc != y.hard_dc || x.decompose_dc != y.decompose_dc || x.adder_size != y.adder_size ||
        x.carry_size != y.carry_size || x.search_all != y.search_all || x.method0 != y.method0 || x.method1 != y.method1 ||
        x.qints.size() != y.qints.size() || x.lats.size() != y.lats.size())
        return false;
    // bitwise comparisons: anything that is not literally the same input is simply solved again
    if (!x.qints.empty() && std::memcmp(x.qints.data(), y.qints.data(), sizeof(QInt) * x.qints.size()) != 0) return false;
    if (!x.lats.empty() && std::memcmp(x.lats.data(), y.lats.data(), sizeof(float) * x.lats.size()) != 0) return false;
    return a.kernel == b.kernel || std::memcmp(a.kernel, b.kernel, sizeof(float) * (size_t)a.n_in * a.n_out) == 0;
}
uint64_t problem_hash(const Problem &p) {
    uint64_t h = 1469598103934665603ull;
    // (every hashed array is a multiple of 4 bytes.)  Four independent accumulators over 32-byte strides: the multiply chain of
    // a single one made hashing the 64 matrices of a batch 5 ms of serial host time; the value only routes the de-duplication,
    // equal hashes are settled by same_problem()
    auto mix = [&](const void *data, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(data);
        uint64_t a[4] = {h, h ^ 0x9E3779B97F4A7C15ull, h ^ 0xC2B2AE3D27D4EB4Full, h ^ 0x165667B19E3779F9ull};
        size_t i = 0;
        for (; i + 32 <= n; i += 32)
            for (int q = 0; q < 4; ++q) {
                uint64_t w;
                std::memcpy(&w, b + i + 8 * q, 8);
                a[q] = (a[q] ^ w) * 1099511628211ull;
                a[q] ^= a[q] >> 29;
            }
        for (; i + 4 <= n; i += 4) {
            uint32_t w;
            std::memcpy(&w, b + i, 4);
            a[0] = (a[0] ^ w) * 1099511628211ull;
            a[0] ^= a[0] >> 29;
        }
        h = a[0];
        for (int q = 1; q < 4; ++q) {
            h = (h ^ a[q]) * 1099511628211ull;
            h ^= h >> 29;
        }
    };
    mix(&p.n_in, sizeof p.n_in);
    mix(&p.n_out, sizeof p.n_out);
    mix(p.kernel, sizeof(float) * (size_t)p.n_in * p.n_out);
    if (!p.opt.qints.empty()) mix(p.opt.qints.data(), sizeof(QInt) * p.opt.qints.size());
    if (!p.opt.lats.empty()) mix(p.opt.lats.data(), sizeof(float) * p.opt.lats.size());
    return h;
}
}  // namespace

// Identical problems of a batch (same matrix, options, intervals and latencies -- the tracer's loop over the row
// vectors of a traced tensor, reference trace/fixed_variable_array.py:368-371, produces many) are solved once and the
// result is copied.  The solver is deterministic, so this cannot change any result.
std::vector<PipeResult> solve_batch(Backend &be, const std::vector<Problem> &problems, std::vector<ChainStats> *stats) {
    const size_t n = problems.size();
    std::vector<int> rep(n);
    std::vector<Problem> uniq;
    std::unordered_multimap<uint64_t, int> seen;
    for (size_t i = 0; i < n; ++i) {
        if (problems[i].n_in <= 0 || problems[i].n_out <= 0 || !problems[i].kernel) throw std::invalid_argument("kernel must be a non-empty 2-D matrix");
        const uint64_t h = problem_hash(problems[i]);
        int found = -1;
        auto range = seen.equal_range(h);
        for (auto it = range.first; it != range.second && found < 0; ++it)
            if (same_problem(uniq[it->second], problems[i])) found = it->second;
        if (found < 0) {
            found = (int)uniq.size();
            uniq.push_back(problems[i]);
            seen.emplace(h, found);
        }
        rep[i] = found;
    }
    if (uniq.size() == n) return solve_batch_unique(be, problems, stats);
    std::vector<ChainStats> ustats;
    std::vector<PipeResult> ures = solve_batch_unique(be, uniq, &ustats);
    std::vector<PipeResult> res(n);
    if (stats) stats->assign(n, ChainStats{});
    for (size_t i = n; i-- > 0;) {  // the first occurrence (visited last) takes the original, later ones copies
        bool first = true;
        for (size_t j = 0; j < i && first; ++j) first = rep[j] != rep[i];
        if (first) {
            res[i] = std::move(ures[rep[i]]);
            if (stats && rep[i] < (int)ustats.size()) (*stats)[i] = ustats[rep[i]];  // work is attributed to the first occurrence only
        } else
            res[i] = ures[rep[i]];
    }
    return res;
}

namespace {
std::vector<PipeResult> solve_batch_unique(Backend &be, const std::vector<Problem> &problems, std::vector<ChainStats> *stats) {
    const bool verbose = std::getenv("DA4ML_HIP_VERBOSE") != nullptr;
    auto lap = [verbose, last = std::chrono::steady_clock::now()](const char *what) mutable {
        auto now = std::chrono::steady_clock::now();
        if (verbose) std::fprintf(stderr, "[da4ml_hip] host: %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    };
    std::vector<ProblemState> ps(problems.size());
    std::vector<Candidate> cands;
    std::vector<size_t> fresh;  // problems whose matrix is seen for the first time
    for (size_t i = 0; i < problems.size(); ++i) {
        const Problem &p = problems[i];
        ProblemState &s = ps[i];
        s.p = &p;
        s.qints = p.opt.qints.empty() ? std::vector<QInt>(p.n_in, QInt{-128.0f, 127.0f, 1.0f}) : p.opt.qints;
        s.lats = p.opt.lats.empty() ? std::vector<float>(p.n_in, 0.0f) : p.opt.lats;
        if ((int)s.qints.size() != p.n_in || (int)s.lats.size() != p.n_in)
            throw std::invalid_argument("qintervals / latencies must have one entry per kernel row");
        for (size_t j = 0; j < i && !s.s1p; ++j) {
            const Problem &o = problems[j];
            if (o.n_in == p.n_in && o.n_out == p.n_out &&
                (o.kernel == p.kernel || std::memcmp(o.kernel, p.kernel, sizeof(float) * (size_t)p.n_in * p.n_out) == 0))
                s.s1p = ps[j].s1p;
        }
        if (!s.s1p) {
            s.s1p = std::make_shared<Stage1>();
            fresh.push_back(i);  // centred below, all new matrices on the host threads together
        }
        int log2_n = (int)std::ceil(std::log2((float)p.n_in));
        std::vector<std::pair<int, int>> tries;  // (hard_dc, decompose_dc)
        if (!p.opt.search_all)
            tries.emplace_back(p.opt.hard_dc, p.opt.decompose_dc);
        else {
            int hdc = p.opt.hard_dc < 0 ? 1000000000 : p.opt.hard_dc;
            for (int d = -1; d <= std::min(hdc, log2_n); ++d) tries.emplace_back(hdc, d);
        }
        for (auto [hdc, ddc] : tries) {
            Candidate c;
            c.problem = (int)i;
            c.method0 = p.opt.method0;
            c.method1 = p.opt.method1;
            c.hard_dc = hdc;
            if (c.method1 == "auto") c.method1 = (hdc >= 6 || has_dc_suffix(c.method0)) ? c.method0 : c.method0 + "-dc";
            if (hdc == 0 && !has_dc_suffix(c.method0)) c.method0 += "-dc";
            c.decompose_dc = ddc == -2 ? std::min(hdc, log2_n) : std::min({hdc, ddc, log2_n});
            c.phase = hdc >= 0 ? Candidate::NEED_MINLAT : Candidate::NEED_STAGE0;
            s.cand.push_back((int)cands.size());
            cands.push_back(std::move(c));
        }
    }

    parallel_for(fresh.size(), [&](size_t k) { stage1_prepare(*ps[fresh[k]].s1p, problems[fresh[k]].kernel, problems[fresh[k]].n_in, problems[fresh[k]].n_out); });
    lap("problem set-up");
    auto prepare_stage0 = [&](Candidate &c) {
        ProblemState &s = ps[c.problem];
        if (c.decompose_dc < 0 && c.hard_dc >= 0) c.method0 = c.method1 = (c.method0 != "dummy") ? "wmc-dc" : "dummy";
        stage1_split(be, *s.s1p, c.decompose_dc, c.m0, c.m1);
    };

    struct Pending {
        int cand;
        int what;  // 0 = minimal latency probe, 1 = stage 0, 2 = stage 1
        int problem;
    };
    while (true) {
        std::vector<ChainJob> jobs;
        std::vector<Pending> owners;
        std::vector<char> minlat_queued(ps.size(), 0);
        // stage-1 work of this round (distance matrix once per problem on the device, MST + m0/m1 per candidate on
        // host threads)
        {
            std::vector<size_t> todo;
            for (size_t ci = 0; ci < cands.size(); ++ci) {
                Candidate &c = cands[ci];
                ProblemState &s = ps[c.problem];
                bool ready = c.phase == Candidate::NEED_STAGE0 || (c.phase == Candidate::NEED_MINLAT && s.minlat_known);
                if (!ready) continue;
                todo.push_back(ci);
                int ddc = c.decompose_dc;
                if (ddc != -1) stage1_distances(be, *s.s1p);
            }
            parallel_for(todo.size(), [&](size_t k) {
                Candidate &c = cands[todo[k]];
                ProblemState &s = ps[c.problem];
                if (c.phase == Candidate::NEED_MINLAT) {
                    c.allowed = c.hard_dc + s.minlat;
                    c.phase = Candidate::NEED_STAGE0;
                }
                prepare_stage0(c);
            });
        }
        for (size_t ci = 0; ci < cands.size(); ++ci) {
            Candidate &c = cands[ci];
            ProblemState &s = ps[c.problem];
            const Problem &p = *s.p;
            if (c.phase == Candidate::NEED_MINLAT) {
                if (s.minlat_known) {
                    c.allowed = c.hard_dc + s.minlat;
                    c.phase = Candidate::NEED_STAGE0;
                } else if (!minlat_queued[c.problem]) {
                    // minimal_latency(): adder trees straight from the CSD digits (api.cc:11-26)
                    jobs.push_back(ChainJob{p.kernel, p.n_in, p.n_out, M_DUMMY, s.qints.data(), s.lats.data(), p.opt.adder_size, p.opt.carry_size});
                    owners.push_back(Pending{(int)ci, 0, c.problem});
                    minlat_queued[c.problem] = 1;
                    continue;
                } else
                    continue;
            }
            if (c.phase == Candidate::NEED_STAGE0) {
                jobs.push_back(ChainJob{c.m0.data(), p.n_in, p.n_out, parse_method(c.method0), s.qints.data(), s.lats.data(), p.opt.adder_size, p.opt.carry_size});
                owners.push_back(Pending{(int)ci, 1, c.problem});
            } else if (c.phase == Candidate::NEED_STAGE1) {
                jobs.push_back(ChainJob{c.m1.data(), p.n_out, p.n_out, parse_method(c.method1), c.q_mid.data(), c.lat_mid.data(), p.opt.adder_size, p.opt.carry_size});
                owners.push_back(Pending{(int)ci, 2, c.problem});
            }
        }
        if (jobs.empty()) break;
        lap("stage 1 + job list");
        std::vector<ChainOut> outs(jobs.size());
        auto t_rc = std::chrono::steady_clock::now();
        {  // chains without a greedy loop are finished on the host; the others go to the backend together
            std::vector<uint8_t> on_host(jobs.size(), 0);
            parallel_for(jobs.size(), [&](size_t k) { on_host[k] = host_chain(jobs[k], outs[k]) ? 1 : 0; });
            std::vector<ChainJob> dev_jobs;
            std::vector<size_t> dev_idx;
            for (size_t k = 0; k < jobs.size(); ++k)
                if (!on_host[k]) {
                    dev_jobs.push_back(jobs[k]);
                    dev_idx.push_back(k);
                }
            if (!dev_jobs.empty()) {
                std::vector<ChainOut> dev_outs(dev_jobs.size());
                be.run_chains(dev_jobs.data(), dev_outs.data(), (int)dev_jobs.size());
                for (size_t k = 0; k < dev_idx.size(); ++k) outs[dev_idx[k]] = std::move(dev_outs[k]);
            }
        }
        auto t_fin = std::chrono::steady_clock::now();
        for (size_t k = 0; k < jobs.size(); ++k)
            if (outs[k].unknown_method_hit) {
                Candidate &c = cands[owners[k].cand];
                throw std::runtime_error("Unknown method: " + (owners[k].what == 2 ? c.method1 : c.method0));
            }
        std::vector<StageResult> sols(jobs.size());
        // Adder trees in two loops over the host threads: (1) per chain, the op records of the inputs and the greedy picks
        // (sequential inside a chain) and the tree sizes; (2) per (chain, range of columns), the trees -- 1024 pieces for the
        // 64 chains of the benchmark, balanced over all threads of the pool with no thread created on the way
        std::vector<std::vector<int64_t>> first_op(jobs.size());
        parallel_for(jobs.size(), [&](size_t k) { sols[k] = finalize_prepare(jobs[k], outs[k], first_op[k]); });
        size_t chunk_from = 4096;  // digits of a chain from which its columns are split into ranges
        int chunks_per_chain = 16;
        if (const char *e = std::getenv("DA4ML_HIP_TREE_THREADS")) {  // test hook: split small chains too, 4 ranges per "thread"
            chunks_per_chain = std::max(1, std::atoi(e)) * 4;
            chunk_from = 0;
        }
        struct Piece {
            int chain, j0, j1;
        };
        std::vector<Piece> pieces;
        for (size_t k = 0; k < jobs.size(); ++k) {
            const int n_out = jobs[k].n_out;
            const int parts = outs[k].dig_row.size() >= chunk_from ? std::max(1, std::min(n_out, chunks_per_chain)) : 1;
            for (int c = 0; c < parts; ++c)
                pieces.push_back(Piece{(int)k, (int)((long long)n_out * c / parts), (int)((long long)n_out * (c + 1) / parts)});
        }
        parallel_for(pieces.size(), [&](size_t i) {
            const Piece &pc = pieces[i];
            finalize_columns(jobs[pc.chain], outs[pc.chain], sols[pc.chain], first_op[pc.chain], pc.j0, pc.j1);
        });
        if (std::getenv("DA4ML_HIP_VERBOSE"))
            std::fprintf(stderr, "[da4ml_hip] round of %zu chains: run_chains %.2f ms, adder trees %.2f ms\n", jobs.size(),
                         std::chrono::duration<double, std::milli>(t_fin - t_rc).count(),
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fin).count());
        lap("chains + adder trees");
        for (size_t k = 0; k < jobs.size(); ++k) {
            Candidate &c = cands[owners[k].cand];
            ProblemState &s = ps[c.problem];
            if (stats) {
                if (stats->size() < problems.size()) stats->resize(problems.size());
                ChainStats &a = (*stats)[c.problem];
                const ChainStats &b = outs[k].stats;
                a.iterations += b.iterations;
                a.digits0 += b.digits0;
                a.blocks0 += b.blocks0;
                a.rebuilds += b.rebuilds;
                a.table_peak = std::max(a.table_peak, b.table_peak);
                a.scan_slots += b.scan_slots;
                a.partners += b.partners;
                a.matches += b.matches;
            }
            StageResult sol = std::move(sols[k]);
            bool both_wmc_dc = c.method0 == "wmc-dc" && c.method1 == "wmc-dc";
            if (owners[k].what == 0) {
                s.minlat = max_out_latency(sol);
                s.minlat_known = true;
                continue;  // candidates pick the value up at the top of the next round
            }
            if (owners[k].what == 1) {
                c.sol0 = std::move(sol);
                c.q_mid.clear();
                c.lat_mid.clear();
                for (auto idx : c.sol0.out_idxs) {
                    c.lat_mid.push_back(idx >= 0 ? c.sol0.ops[idx].latency : 0.0f);
                    c.q_mid.push_back(idx >= 0 ? c.sol0.ops[idx].q : QInt{0.0f, 0.0f, std::numeric_limits<float>::infinity()});
                }
                if (max_out_latency(c.sol0) > c.allowed && (!both_wmc_dc || c.decompose_dc >= 0)) {
                    c.decompose_dc--;
                    c.phase = Candidate::NEED_STAGE0;
                } else
                    c.phase = Candidate::NEED_STAGE1;
            } else {
                c.sol1 = std::move(sol);
                if (max_out_latency(c.sol1) > c.allowed && (!both_wmc_dc || c.decompose_dc >= 0)) {
                    c.decompose_dc--;
                    c.phase = Candidate::NEED_STAGE0;
                } else
                    c.phase = Candidate::DONE;
            }
        }
        lap("results taken over");
        // the round's device outputs (digits, picks, column lists: megabytes per chain) are released on the pool, not one after the other at the closing brace
        parallel_for(outs.size(), [&](size_t k) {
            outs[k] = ChainOut();
            std::vector<int64_t>().swap(first_op[k]);
        });
        lap("round buffers released");
    }

    lap("candidate bookkeeping");
    std::vector<PipeResult> results(problems.size());
    for (size_t i = 0; i < problems.size(); ++i) {
        ProblemState &s = ps[i];
        int best = 0;
        if (s.p->opt.search_all) {
            std::vector<float> costs;
            for (int ci : s.cand) {
                float total = 0.0f;
                for (const StageResult *st : {&cands[ci].sol0, &cands[ci].sol1})
                    for (const OpRec &op : st->ops) total += op.cost;
                costs.push_back(total);
            }
            for (size_t k = 1; k < costs.size(); ++k)
                if (costs[k] < costs[best]) best = (int)k;
            results[i].picked = best;
        }
        Candidate &w = cands[s.cand[best]];
        results[i].stages.push_back(std::move(w.sol0));
        results[i].stages.push_back(std::move(w.sol1));
    }
    for (Candidate &c : cands) {  // the op lists of the candidates that lost (the winners' were moved out: nothing left to keep)
        recycle_op_list(std::move(c.sol0.ops));
        recycle_op_list(std::move(c.sol1.ops));
    }
    lap("winner selection");
    return results;
}

}  // namespace

}  // namespace da
