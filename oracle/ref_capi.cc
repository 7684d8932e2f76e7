// ref_capi.cc -- C ABI over the REAL reference C++ API (api.hh / mat_decompose.hh / bit_decompose.hh /
// state_opr.hh / indexers.hh of /root/reference/src/da4ml/_binary/cmvm), compiled against oracle/shim.
// Same entry-point shapes as cmvm_oracle.cc (prefix ref_ instead of orc_).  Test infrastructure only.
#include "api.hh"
#include "bit_decompose.hh"
#include "indexers.hh"
#include "mat_decompose.hh"
#include "state_opr.hh"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static thread_local std::string g_err;

static xt::xarray<float> as_kernel(const float *k, int64_t n_in, int64_t n_out) {
    xt::xarray<float> a(xt::shape_t{(size_t)n_in, (size_t)n_out});
    std::memcpy(a.data(), k, sizeof(float) * (size_t)(n_in * n_out));
    return a;
}

struct RefResult {
    PipelineResult pipe;
};

extern "C" {
const char *ref_last_error() { return g_err.c_str(); }
int ref_get_lsb_loc(float x) { return get_lsb_loc(x); }
int ref_iceil_log2(float x) { return iceil_log2(x); }
void ref_cost_add(const float *q0, const float *q1, int64_t shift, int sub, int adder_size, int carry_size, float *out2) {
    auto r = cost_add(QInterval{q0[0], q0[1], q0[2]}, QInterval{q1[0], q1[1], q1[2]}, shift, sub != 0, adder_size, carry_size);
    out2[0] = r.first;
    out2[1] = r.second;
}
int ref_int_arr_to_csd(const int32_t *x, int64_t n, int8_t *out) {
    xt::xarray<int32_t> a(xt::shape_t{(size_t)n});
    std::memcpy(a.data(), x, sizeof(int32_t) * (size_t)n);
    auto csd = _volatile_int_arr_to_csd(a);
    int N = (int)csd.shape(csd.dimension() - 1);
    if (out) std::memcpy(out, csd.data(), csd.size());
    return N;
}
int ref_csd_decompose(const float *kernel, int64_t n_in, int64_t n_out, int do_center, int8_t *csd, int8_t *s0, int8_t *s1) {
    auto k = as_kernel(kernel, n_in, n_out);
    auto [c, a, b] = csd_decompose(k, do_center != 0);
    if (csd) std::memcpy(csd, c.data(), c.size());
    if (s0) std::memcpy(s0, a.data(), a.size());
    if (s1) std::memcpy(s1, b.data(), b.size());
    return (int)c.shape(2);
}
void ref_kernel_decompose(const float *kernel, int64_t n_in, int64_t n_out, int dc, float *m0, float *m1) {
    auto [a, b] = kernel_decompose(as_kernel(kernel, n_in, n_out), dc);
    std::memcpy(m0, a.data(), a.size() * 4);
    std::memcpy(m1, b.data(), b.size() * 4);
}
void *ref_solve(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                int decompose_dc, const float *qints3, const float *lats, int adder_size, int carry_size, int search_all) {
    try {
        std::vector<QInterval> q;
        std::vector<float> l;
        if (qints3)
            for (int64_t i = 0; i < n_in; ++i) q.push_back(QInterval{qints3[3 * i], qints3[3 * i + 1], qints3[3 * i + 2]});
        if (lats) l.assign(lats, lats + n_in);
        auto *r = new RefResult;
        r->pipe = solve(as_kernel(kernel, n_in, n_out), method0, method1, hard_dc, decompose_dc, q, l, adder_size, carry_size,
                        search_all != 0);
        return r;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
// Bounded CPU timing sample with the reference's own functions (create_state, idx_*, update_state): state + pair
// table creation, then greedy iterations until `budget_s` seconds are spent.  Same contract as orc_sample_chain.
int ref_sample_chain(const float *kernel, int64_t n_in, int64_t n_out, const char *method, double budget_s, double *out) {
    try {
        std::vector<QInterval> q(n_in, QInterval{-128.0f, 127.0f, 1.0f});
        std::vector<float> l(n_in, 0.0f);
        std::string m(method);
        auto t0 = std::chrono::steady_clock::now();
        DAState s = create_state(as_kernel(kernel, n_in, n_out), q, l);
        auto t1 = std::chrono::steady_clock::now();
        out[0] = std::chrono::duration<double>(t1 - t0).count();
        out[1] = out[2] = out[3] = 0;
        // optional time marks "iteration seconds-since-start" (env REF_TRACE=<file>): the calibration curve of
        // bench.py's cpu_baseline (tests/golden/make_cpu_calibration.py); instrumentation only
        FILE *trace = nullptr;
        if (const char *tp = std::getenv("REF_TRACE")) trace = std::fopen(tp, "w");
        if (trace) std::fprintf(trace, "0 %.6f\n", out[0]);
        long long next_mark = 1;
        while (true) {
            if (s.freq_stat.empty()) {
                out[3] = 1;
                break;
            }
            Pair pick = m == "mc" ? idx_mc(s) : m == "wmc" ? idx_wmc(s) : m == "wmc-dc" ? idx_wmc_dc(s, true) : idx_mc_dc(s, true);
            if (pick.id0 == -1 || pick.id1 == -1) {
                out[3] = 1;
                break;
            }
            update_state(s, pick, -1, -1);
            out[1] += 1;
            out[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            if (trace && (long long)out[1] >= next_mark) {
                std::fprintf(trace, "%lld %.6f\n", (long long)out[1], out[0] + out[2]);
                std::fflush(trace);
                next_mark = next_mark < 64 ? next_mark * 2 : next_mark + (next_mark < 1024 ? 64 : 256);
            }
            if (out[2] >= budget_s) break;
        }
        if (trace) {
            std::fprintf(trace, "%lld %.6f\n", (long long)out[1], out[0] + out[2]);
            std::fclose(trace);
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int ref_n_stages(void *h) { return (int)((RefResult *)h)->pipe.solutions.size(); }
int ref_picked(void *) { return -1; }
void ref_stage_info(void *h, int s, int64_t *info) {
    const CombLogicResult &st = ((RefResult *)h)->pipe.solutions[s];
    info[0] = st.shape.first;
    info[1] = st.shape.second;
    info[2] = (int64_t)st.ops.size();
    info[3] = st.carry_size;
    info[4] = st.adder_size;
}
void ref_stage_copy(void *h, int s, int64_t *inp_shifts, int64_t *out_idxs, int64_t *out_shifts, int64_t *out_negs,
                    int64_t *ops_i, float *ops_f) {
    const CombLogicResult &st = ((RefResult *)h)->pipe.solutions[s];
    std::memcpy(inp_shifts, st.inp_shifts.data(), st.inp_shifts.size() * 8);
    std::memcpy(out_idxs, st.out_idxs.data(), st.out_idxs.size() * 8);
    std::memcpy(out_shifts, st.out_shifts.data(), st.out_shifts.size() * 8);
    std::memcpy(out_negs, st.out_negs.data(), st.out_negs.size() * 8);
    for (size_t k = 0; k < st.ops.size(); ++k) {
        const Op &o = st.ops[k];
        ops_i[4 * k] = o.id0;
        ops_i[4 * k + 1] = o.id1;
        ops_i[4 * k + 2] = o.opcode;
        ops_i[4 * k + 3] = o.data;
        ops_f[5 * k] = o.qint.min;
        ops_f[5 * k + 1] = o.qint.max;
        ops_f[5 * k + 2] = o.qint.step;
        ops_f[5 * k + 3] = o.latency;
        ops_f[5 * k + 4] = o.cost;
    }
}
void ref_free(void *h) { delete (RefResult *)h; }
}
