// dais_ref_capi.cc -- TEST INFRASTRUCTURE: C entry point over the REFERENCE's own DAIS interpreter
// (/root/reference/src/da4ml/_binary/dais/DAISInterpreter.{hh,cc}, compiled where it lies by oracle/Makefile into
// oracle/_ref/libdais_ref.so).  Used only to pin da4ml_amd/csrc/dais_interp.cc and to generate tests/golden/dais_golden.
#include <cstdint>
#include <span>
#include <string>
#include <vector>

#include "DAISInterpreter.hh"

static thread_local std::string g_err;

extern "C" {
const char *dref_last_error() { return g_err.c_str(); }
// same contract as da_dais_run (single thread): 0 on success, -1 on a C++ exception of the reference
int dref_run(const int32_t *program, int64_t n_words, const double *inputs, int64_t n_samples, double *outputs) {
    try {
        dais::DAISInterpreter interp;
        interp.load_from_binary(std::span<const int32_t>(program, (size_t)n_words));
        const int64_t n_in = program[2], n_out = program[3];
        for (int64_t s = 0; s < n_samples; ++s) {
            std::span<const double> in(inputs + s * n_in, (size_t)n_in);
            std::span<double> out(outputs + s * n_out, (size_t)n_out);
            interp.inference(in, out);
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}
}
