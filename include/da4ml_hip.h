/* da4ml_hip.h -- C ABI of libda4ml_hip.so, the MI355X-native drop-in for da4ml's native CMVM module.
 *
 * Every entry point replaces one binding of the reference's nanobind module `cmvm_bin`
 * (reference src/da4ml/_binary/cmvm/bindings.cc:227-264, re-exported by src/da4ml/_binary/__init__.py:4 and
 * src/da4ml/cmvm/__init__.py:7).  Plain pointers and sizes only; no Python, torch or HIP types cross this
 * boundary.  All matrices are dense row-major; `kernel` is float32 [n_in, n_out] holding dyadic rationals
 * exactly (the reference's own contract, trace/fixed_variable_array.py:76).
 *
 * Error handling: functions returning int give 0 on success and a negative DA_ERR_* otherwise; functions returning
 * a handle give NULL on error.  da_last_error() returns the message of the calling thread's last failure
 * (the reference raises C++ exceptions that nanobind maps to Python exceptions, bindings.cc / api.cc:207-240).
 * The library never keeps pointers to caller memory after a call returns.
 *
 * Threading: calls are serialised per device by an internal mutex; solves of a batch run concurrently on the GPU.
 */
#ifndef DA4ML_HIP_H
#define DA4ML_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DA_OK 0
#define DA_ERR_RUNTIME (-1)  /* maps to Python RuntimeError (std::runtime_error in the reference) */
#define DA_ERR_VALUE (-2)    /* maps to Python ValueError   (std::invalid_argument in the reference) */
#define DA_ERR_NO_DEVICE (-3)

typedef struct da_result da_result; /* one solved Pipeline (two CombLogic stages) */

/* ---- library / device ------------------------------------------------------------------------------------ */
const char *da_last_error(void);
/* DA_ERR_* class of the calling thread's last failure -- which Python exception the reference would have raised
 * (std::invalid_argument -> ValueError, std::runtime_error -> RuntimeError; bindings.cc / state_opr.cc:70-72).  Needed
 * by the handle-returning entry points (da_solve returns NULL without a code). */
int da_last_error_code(void);
const char *da_version(void);
int da_device_count(void);     /* number of visible HIP devices (0 if none) */
int da_set_device(int device); /* device used by subsequent calls of this process (default 0) */

/* ---- scalar helpers -------------------------------------------------------------------------------------- */
/* cmvm_bin.get_lsb_loc (bindings.cc:229 -> bit_decompose.cc:10-20) */
int da_get_lsb_loc(float x);
/* cmvm_bin.iceil_log2 (bindings.cc:230 -> indexers.hh:12-18) */
int da_iceil_log2(float x);
/* cmvm_bin.cost_add (bindings.cc:25-41,249-263 -> state_opr.cc:31-67); q0,q1 = {min,max,step}; out2 = {latency,cost} */
int da_cost_add(const float *q0, const float *q1, int64_t shift, int sub, int adder_size, int carry_size, float *out2);

/* ---- decompositions (run on the GPU) --------------------------------------------------------------------- */
/* cmvm_bin.int_arr_to_csd (bindings.cc:43-61,228 -> bit_decompose.cc:22-42).  Returns the digit count N (>= 1) or a
 * negative error; `out` may be NULL to query N, otherwise int8 [n, N]. */
int da_int_arr_to_csd(const int32_t *x, int64_t n, int8_t *out);
/* cmvm_bin.csd_decompose (bindings.cc:63-103,231 -> bit_decompose.cc:45-62).  Returns N or a negative error; `csd`
 * (int8 [n_in, n_out, N]), `shift0` (int8 [n_in]), `shift1` (int8 [n_out]) may be NULL to query N. */
int da_csd_decompose(const float *kernel, int64_t n_in, int64_t n_out, int center, int8_t *csd, int8_t *shift0, int8_t *shift1);
/* cmvm_bin.kernel_decompose (bindings.cc:232-234 -> mat_decompose.cc:63-169): m0 float32 [n_in,n_out], m1 [n_out,n_out] */
int da_kernel_decompose(const float *kernel, int64_t n_in, int64_t n_out, int dc, float *m0, float *m1);

/* ---- solve ------------------------------------------------------------------------------------------------ */
/* cmvm_bin.solve (bindings.cc:184-225,235-248 -> api.cc:147-250).  `qintervals` is float32 [n_in,3] or NULL
 * (default (-128,127,1)), `latencies` float32 [n_in] or NULL (default 0).  Steps must be positive normal numbers (DA_ERR_VALUE
 * otherwise); they need not be powers of two (the latency model reads -log2f of any other step from a table the host builds with
 * its own libm, one row per distinct mantissa). */
da_result *da_solve(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                    int decompose_dc, const float *qintervals, const float *latencies, int adder_size, int carry_size,
                    int search_all_decompose_dc);

/* Batched form of da_solve (addition): `count` independent problems sharing the option
 * set are solved concurrently on the device.  kernels[i] is [n_in[i], n_out[i]]; qintervals / latencies may be NULL
 * or hold per-problem pointers (each possibly NULL).  results[i] receives a handle (to be freed with da_free).
 * Returns DA_OK, or an error in which case no handle is returned. */
int da_solve_batch(int count, const float *const *kernels, const int64_t *n_in, const int64_t *n_out, const char *method0,
                   const char *method1, int hard_dc, int decompose_dc, const float *const *qintervals,
                   const float *const *latencies, int adder_size, int carry_size, int search_all_decompose_dc,
                   da_result **results);

/* ---- column-sharded solve (BASELINE config C4; addition, no reference counterpart) ------------------------------------------
 * da_solve with every greedy chain sharded over the output COLUMNS of its matrix across `world` processes (one per GPU):
 * rank g holds the digits of columns [g n_out / world, (g+1) n_out / world), the pair-count table is replicated, and per
 * greedy step the ranks exchange two int32 slabs by all-reduce(sum) -- the loops it shards are column-outermost in the
 * reference (state_opr.cc:117,249,307; adder trees cmvm_core.cc:103).  Every rank calls this with identical arguments and
 * receives the identical, complete result (= da_solve's).  The library does not link a communication library: the host
 * process supplies the collective (RCCL through torch.distributed in da4ml_amd.multi_gpu.solve_column_sharded).
 * `allreduce(ctx, buf, count, on_device)`: in-place sum of `count` int32 at `buf` over all ranks; `buf` is device memory
 * of the current device when on_device != 0 (the producing kernels have completed), host memory otherwise; returns when
 * the result is in place.  stats3 (may be NULL) = {sharded chains, greedy steps, all-reduce calls}.  Latency-bound by
 * construction (two collectives per greedy step); the layout that scales is one matrix per rank (da_solve_batch). */
typedef void (*da_allreduce_i32)(void *ctx, void *buf, int64_t count, int on_device);
da_result *da_solve_sharded(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                            int decompose_dc, const float *qintervals, const float *latencies, int adder_size, int carry_size,
                            int search_all_decompose_dc, int rank, int world, da_allreduce_i32 allreduce, void *ctx, int64_t *stats3);
/* The same with the library's own RCCL transport (csrc/cmvm_rccl.*): `ncclAllReduce` in place on the library's HIP stream,
 * stream-ordered with the kernels that produce and consume the exchange buffers -- no callback, no host synchronisation per
 * exchange.  librccl.so is opened at run time (no link-time dependency).  Rank 0 obtains the 128-byte unique id
 * (da_rccl_unique_id = ncclGetUniqueId) and hands it to every rank by its own means (da4ml_amd.multi_gpu broadcasts it through
 * torch.distributed); every rank then calls da_solve_sharded_rccl with identical arguments.  The communicator of an id is kept
 * for further calls -- callers hand the SAME id to every solve of a process group (a fresh id per call would build, and keep, a
 * new communicator each time); da_rccl_shutdown destroys the kept communicators (ncclCommDestroy; every rank calls it, no solve
 * running, the process group still alive) and returns their number, -1 on error. */
int da_rccl_unique_id(void *id128);
int da_rccl_shutdown(void);
da_result *da_solve_sharded_rccl(const float *kernel, int64_t n_in, int64_t n_out, const char *method0, const char *method1, int hard_dc,
                                 int decompose_dc, const float *qintervals, const float *latencies, int adder_size, int carry_size,
                                 int search_all_decompose_dc, int rank, int world, const void *id128, int64_t *stats3);
/* To be called by the all-reduce callback's owner when a collective failed (the callback returns nothing and must not unwind
 * through the library): the running da_solve_sharded stops at the next exchange and fails with a runtime error instead of
 * continuing with a buffer that was not reduced. */
void da_comm_abort(void);
/* int32 elements handed to all-reduce(sum) by the last da_solve_sharded / da_solve_sharded_rccl of this process (x 4 = bytes per rank and
 * exchange direction): the exchange volume of a column-sharded solve, for bench.py's c4 workload. */
int64_t da_shard_exchanged_elements(void);

/* ---- result access (da4ml.types.Pipeline / CombLogic / Op, bindings.cc:106-151) ---------------------------- */
int da_n_stages(const da_result *r);
/* index of the winning decompose_dc candidate when search_all_decompose_dc was set, else -1 */
int da_picked(const da_result *r);
/* info[0..4] = n_in, n_out, n_ops, carry_size, adder_size */
int da_stage_info(const da_result *r, int stage, int64_t *info);
/* inp_shifts[n_in], out_idxs[n_out], out_shifts[n_out], out_negs[n_out] (0/1), ops_i[n_ops,4] = id0,id1,opcode,data,
 * ops_f[n_ops,5] = qint.min,qint.max,qint.step,latency,cost */
int da_stage_copy(const da_result *r, int stage, int64_t *inp_shifts, int64_t *out_idxs, int64_t *out_shifts, int64_t *out_negs,
                  int64_t *ops_i, float *ops_f);
/* stats[0..7] = greedy iterations, initial digits, -, selection rounds, peak pair blocks, re-read table slots,
 * partner rows updated, substituted digits -- summed over the chains run for this problem.  "Peak pair blocks" follows the
 * table only when the environment variable DA4ML_HIP_STATS=1 is set at the call (the update kernel then tallies the blocks it
 * creates and deletes: instrumentation, 1.8 % of a greedy step); without it the figure is the initial block count. */
int da_result_stats(const da_result *r, int64_t *stats);
/* Releases a result handle (NULL is ignored).  The op lists of the result are kept by the library for the next solve
 * (at most 1 GiB in total) instead of being returned to the allocator. */
void da_free(da_result *r);

/* ---- DAIS program executor (host) -------------------------------------------------------------------------- */
/* dais_bin.run_interp (reference src/da4ml/_binary/dais/bindings.cc:30-131 -> DAISInterpreter.cc:10-388; program layout
 * docs/dais.md:70-95 = CombLogic.to_binary()).  Integer-exact execution of `program` (int32 [n_words]) on `n_samples`
 * input rows (float64 [n_samples, n_in], row-major) into `outputs` (float64 [n_samples, n_out]); n_threads <= 0 = all
 * host threads, samples are split in chunks of at least 32 like the reference.  Runs on the host (the reference's does
 * too): the functional checker of solver results behind CombLogic.predict, not part of the solver path.
 * Returns DA_OK or DA_ERR_RUNTIME (message: da_dais_last_error, calling thread). */
int da_dais_run(const int32_t *program, int64_t n_words, const double *inputs, int64_t n_samples, double *outputs, int n_threads);
const char *da_dais_last_error(void);
/* Same execution with the executor chosen by `where`:
 *   DA_DAIS_HOST        the host block executor above;
 *   DA_DAIS_DEVICE      HIP kernel k_dais_run on the current device: one thread per sample, register file
 *                       [slot][sample] in HBM with liveness-compacted slots (csrc/dais_gpu.hip); fails without a GPU;
 *   DA_DAIS_HOST_SCALAR the device executor's per-value code (csrc/dais_core.h) run sample by sample on the host --
 *                       test aid that pins the shared arithmetic without a GPU.
 * No reference counterpart (the reference's interpreter is host-only); results are identical for all three. */
#define DA_DAIS_HOST 0
#define DA_DAIS_DEVICE 1
#define DA_DAIS_HOST_SCALAR 2
int da_dais_run_on(const int32_t *program, int64_t n_words, const double *inputs, int64_t n_samples, double *outputs, int n_threads,
                   int where);

/* ---- instrumentation for the benchmark harness ------------------------------------------------------------- */
/* t[31] = chains summed over the sampled launches; t[30] = capacity retries; t[18..29] = shader-clock cycles per kernel phase (7 of k_iter_select, 5 of k_iter_update);
 * t[0..17] = loop_ms (HIP events around the greedy-loop launches), dist_ms, total_ms, lockstep iterations,
 * greedy iterations, table groups re-read, partner rows, chains, table bytes, arena bytes, sampled k_iter_select ms,
 * sampled k_iter_update ms, number of samples, count blocks found, count blocks inserted, partner cells read,
 * count-block bytes (2K per touched block), partner-cell bytes -- accumulated since the last reset */
int da_timings(double *t, int reset);  /* (blocks found / created are tallied only under DA4ML_HIP_STATS=1, see da_result_stats) */
/* Further engine counters, same accumulation and reset as da_timings (call BEFORE a resetting da_timings); writes min(n, 16) values:
 * out[0] algorithmic bytes of k_iter_select (device-counted), out[1] host ms spent queueing greedy-loop
 * launches, out[2..15] reserved (0); returns the number written */
int da_engine_stats(double *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* DA4ML_HIP_H */
