// cmvm_host.h -- host side of the MI355X CMVM solver: everything around the greedy chains.
//
// Split of work:
//   device (cmvm_engine.hip)   centring + CSD recoding, pair-count table, the greedy selection /
//                              substitution / recount loop, stage-1 column distance matrix
//   host   (this file)         option resolution, stage-1 minimum spanning tree and m0/m1 assembly,
//                              per-output adder trees, op records (interval / latency / cost with the
//                              host libm), candidate search and arg-min
// The host code talks to the chains through the abstract `Backend`; the product links the HIP backend
// only.  (tests/model/ links a sequential model backend to exercise this file without a GPU.)
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "cmvm_core.h"

namespace da {

struct QInt {
    float lo, hi, step;
};
struct OpRec {  // reference types.hh:14-21
    int64_t id0, id1, opcode, data;
    QInt q;
    float latency, cost;
};
struct StageResult {  // reference types.hh:153-162
    int64_t n_in = 0, n_out = 0;
    std::vector<int64_t> inp_shifts, out_idxs, out_shifts, out_negs;
    std::vector<OpRec> ops;
    int carry_size = -1, adder_size = -1;
};
struct PipeResult {
    std::vector<StageResult> stages;
    int picked = -1;  // index of the winning decompose_dc candidate (search mode), else -1
};

struct ChainStats {
    int64_t iterations = 0, digits0 = 0, blocks0 = 0, rebuilds = 0, table_peak = 0, scan_slots = 0, partners = 0,
            matches = 0;
};

// One greedy chain = cmvm() of the reference (cmvm_core.cc:10-72) on one matrix.
struct ChainJob {
    const float *kernel = nullptr;  // [n_in, n_out] row-major, dyadic values
    int n_in = 0, n_out = 0;
    int method = M_WMC;  // da::Method, or -1 for an unknown method string (errors only if the table is non-empty)
    const QInt *qints = nullptr;  // [n_in]
    const float *lats = nullptr;  // [n_in]
    int adder_size = -1, carry_size = -1;
};
struct ChainOut {
    int error = E_OK;
    bool unknown_method_hit = false;
    int n_bits = 0;
    std::vector<int8_t> shift0, shift1;
    std::vector<int32_t> picks;   // 4 per iteration: id0, id1, sub, shift
    std::vector<float> row_lat;   // latency of every row as computed by the chain (inputs first)
    // surviving digits, column major: for column j the entries [col_start[j], col_start[j+1]) hold
    // (row id ascending, cell) -- the input of the adder-tree stage
    std::vector<uint32_t> col_start, dig_row;
    std::vector<uint64_t> dig_cell;
    ChainStats stats;
};

class Backend {
  public:
    virtual ~Backend() = default;
    // run independent chains to completion
    virtual void run_chains(const ChainJob *jobs, ChainOut *outs, int n) = 0;
    // stage-1 distances: aug is [n_in, W] int32 (column 0 is the zero column);
    // d0[a*W+b] = sum_i nnzCSD(aug[i,a]-aug[i,b]), d1 likewise with '+'
    virtual void column_distances(const int32_t *aug, int n_in, int W, int64_t *d0, int64_t *d1) = 0;
    // centring + CSD of a 2-D float matrix: digits [n_in, n_out, N] (returned N), shifts
    virtual int csd_decompose(const float *kernel, int n_in, int n_out, bool center, std::vector<int8_t> &csd,
                              std::vector<int8_t> &s0, std::vector<int8_t> &s1) = 0;
    // NAF digits of a flat int32 array with the width derived from its global max
    virtual int int_to_csd(const int32_t *x, int64_t n, std::vector<int8_t> &csd) = 0;
};

struct SolveOptions {
    std::string method0 = "wmc", method1 = "auto";
    int hard_dc = -1, decompose_dc = -2;
    std::vector<QInt> qints;   // empty -> (-128, 127, 1)
    std::vector<float> lats;   // empty -> 0
    int adder_size = -1, carry_size = -1;
    bool search_all = true;
};

int parse_method(const std::string &name);  // da::Method or -1

// exact host versions of the reference's scalar helpers (state_opr.cc:8-67)
QInt qint_add(const QInt &a, const QInt &b, int64_t shift, bool neg_a, bool neg_b);
void cost_add(const QInt &a, const QInt &b, int64_t shift, bool sub, int adder_size, int carry_size, float &dlat,
              float &cost);

// host description of the local libm's log2f for the device latency model
Log2Table measure_log2_table();
// -log2f of the non-power-of-two input steps of a chain, by the local libm (StepLog2 in cmvm_core.h): distinct mantissas of
// the inputs' steps (constant-zero inputs skipped: their step is never used) and, per mantissa, 256 values indexed by the
// biased exponent -- one row per distinct mantissa, however many the inputs have.
struct StepLog2Host {
    std::vector<uint32_t> mant;
    std::vector<float> tab;
    void build(const QInt *q, int n_in);
    StepLog2 view() const { return StepLog2{(int)mant.size(), mant.data(), tab.data()}; }
};

// bit_decompose.hh:25-34 on the host (needed for the m0/m1 assembly)
void center_matrix(std::vector<float> &a, int n_in, int n_out, std::vector<int8_t> &s0, std::vector<int8_t> &s1);

// mat_decompose.cc:63-137
void kernel_decompose(Backend &be, const float *kernel, int n_in, int n_out, int dc, std::vector<float> &m0,
                      std::vector<float> &m1);

// cmvm_core.cc:89-225 from a finished chain
// Op lists are megabytes (65 k ops x 56 bytes for a 256x256 chain) and a batch holds 64 of them: handed to the allocator, each
// is an mmap on creation, ~900 page faults on first touch and a munmap (with a TLB shoot-down on every core the process has
// run on) on release -- per result and per call.  Released lists are kept here instead (bounded) and handed to the next call.
std::vector<OpRec> take_op_list(size_t capacity);  // empty vector with at least this capacity (recycled if one fits)
void recycle_op_list(std::vector<OpRec> &&ops);    // called by da_free / when a result is dropped
StageResult finalize_prepare(const ChainJob &job, const ChainOut &out, std::vector<int64_t> &first_op);  // inputs + greedy picks, tree sizes
void finalize_columns(const ChainJob &job, const ChainOut &out, StageResult &r, const std::vector<int64_t> &first_op, int j0, int j1);  // trees of columns [j0, j1)
StageResult finalize_chain(const ChainJob &job, const ChainOut &out, int inner_threads = 1);  // both, on inner_threads threads of its own  // inner_threads: host threads for the per-column trees

// api.cc:147-250 for a batch of independent problems (one entry per matrix); problems progress together so
// that every round submits all currently runnable chains to the backend at once.
struct Problem {
    const float *kernel;
    int n_in, n_out;
    SolveOptions opt;
};
std::vector<PipeResult> solve_batch(Backend &be, const std::vector<Problem> &problems, std::vector<ChainStats> *stats = nullptr);

}  // namespace da
