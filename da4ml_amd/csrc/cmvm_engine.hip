// cmvm_engine.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the CMVM greedy engine and
// the HIP implementation of da::Backend.  No CUDA compatibility layer, no dual paths: this file only
// targets MI355X.
//
// Reference loops replaced:
//   k_prepare        _center + global digit width            bit_decompose.hh:25-34, bit_decompose.cc:22-27
//   k_init_cells     CSD recoding + SparseExpr build          bit_decompose.cc:28-42, state_opr.cc:93-112
//   k_init_pairs     all-pairs enumeration / FreqMap::initialize   state_opr.cc:117-143, types.hh:73-100
//   k_iter_select2   idx_mc / idx_mc_dc / idx_wmc / idx_wmc_dc + update_expr   indexers.cc:6-90, state_opr.cc:227-283
//                    (two workgroups per chain: the search finds the next pick one step AHEAD, beside the substitution of the current one;
//                    ordinary and column-sharded chains alike -- the one-block k_iter_select of rounds 1-5 is gone)
//   k_iter_update    update_stats (purge + regenerate) as an exact incremental update   state_opr.cc:285-345
//   k_extract        digit gather of to_solution             cmvm_core.cc:103-113
//   k_col_dist       stage-1 CSD Hamming distances           mat_decompose.cc:75-93
//
// Data layout in HBM, per chain (all arrays carved from one arena, see HipBackend::run_chains):
//   rlist   row lists: every row is a list of (column, cell) entries sorted by column.  The n_in input rows are DENSE
//           (n_out entries, entry j = column j); every later row is SPARSE: it only ever holds the columns it was
//           created with (digits only disappear from a row), so a typical row is one 128-byte line.  A cell holds the
//           digits of (row, column) as two position bitmasks (cmvm_core.h).  Narrow layout (n_out <= 256, <= 12 digits):
//           one u32 per entry = col:8 | minus:12 | plus:12; wide layout: 16 bytes = u64 cell + u32 col.
//   rowoff  [rcap] {offset, length} of every row's list (bump-allocated in creation order)
//   rows    [rcap]         RowInfo interval + latency of every row
//   collist [n_out][lcap]  u64     rows that have (had) digits in a column, ascending: row:24 | len:12 | off:28 (digit
//                                  gather at the end; the column-sharded chain's flags)
//   colbits [n_out][rcap/32]       bit r of column j: row r has digits in column j NOW; the partner rows of a greedy step are
//                                  the OR of the substituted columns' bitmaps
//   table   C slots (power of two), open addressing on (id0,id1):
//             hkey[C] u64 (16 keys = one 128-byte line = one probe bucket), hrank[C] u32 (selection rank of the block's
//             best key, 0 = none; the array the selection re-reads), hblk[C] one PAYLOAD LINE per slot:
//             {n_overlap, |dlat|, rank copy, index of the best key, K x u16 exact occurrence counts} -- a block update
//             reads and writes ONE line instead of four arrays; hidx[C] u8 behind hrank: the best key's index once more, dense (the search
//             reads rank, key and index of a whole group from three dense arrays)
//   grec    [C / GS]       32 B    per group of GS consecutive slots: bound word (rank << 32 | tie_word >> 23) of its best entry, the highest value an
//                                  update lowered in the group (glow), the best entry's exact tie word, flag "tie word unknown" (GroupRec)
//
// The reference keeps a sorted table of all pairs with count >= 2, purges every entry touching the two
// substituted rows and regenerates their pairs against ALL rows every iteration.  Here the counts are
// maintained exactly by subtracting, for every "partner" row that shares a substituted column, the pair
// occurrences of the consumed digits, and by creating the blocks of the new row -- O(consumed digits x
// column population) instead of O(row digits x all rows) per iteration.  Selection never scans the table:
// it keeps per-group bounds of the rank -- tight ones: an update that lowers a group's best entry leaves a mark (glow) that the next
// selection acts on --, and a step changes only entries of its pick's rows, so the best entry it leaves untouched is the next pick unless
// one of the few entries the update lists beats it (k_iter_select2).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cmvm_core.h"
#include "cmvm_gpu.h"
#include "cmvm_host.h"

namespace da {
namespace gpu {

#define HIP_CHECK(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess)                                                                                    \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);         \
    } while (0)

constexpr uint64_t KEY_EMPTY = ~0ull;
// Tombstones carry the low two bits of the launch that wrote them (KEY_TOMB - id, id = 0..3).  A slot freed in a
// launch is never re-used within the SAME launch: the deleting wave's stores to the slot (counts, rank, tombstone)
// and a re-using wave's stores would otherwise be unordered (different waves, possibly different XCD L2s).  Kernel
// boundaries order everything, so the slot is claimable from the next launch on.
constexpr uint64_t KEY_TOMB = ~0ull - 1;
constexpr uint64_t KEY_TOMB_LO = KEY_TOMB - 3;  // keys >= KEY_TOMB_LO and != KEY_EMPTY are tombstones
constexpr int WAVE = 64;
constexpr int UPD_THREADS = 256;   // k_iter_update block
constexpr int UPD_WAVES = UPD_THREADS / WAVE;
#ifndef DA_MAX_GROUPS
#define DA_MAX_GROUPS 2048  // group records the search sweeps every step (16 waves x 64 lanes x 2).  Measured (MI355X, round 6, C3 batch / one chain, us per step; slots of a
                            // group read held in registers: DA_SEL_RK x 64): 4096 x 512 slots (RK 8) 34.4 / 23.0; 2048 x 1024 (RK 16) 33.1 / 22.2, (RK 8) 40.5 / 25.3; 1024 x 2048 (RK 16) 47.4 / 28.1
#endif
constexpr int MAX_GROUPS = DA_MAX_GROUPS;
#ifndef DA_IDS_LDS
#define DA_IDS_LDS 2048  // partner row ids of a step staged in LDS by the substitution block (tests/emu builds with 16: its problems are small, both paths run)
#endif

// Per-phase shader-clock timers of k_iter_select2 / k_iter_update (tests/gpu_profile.py).  They cost SGPRs, VALU time and --
// every s_memtime is followed by a wait for all outstanding LDS and scalar operations -- latency on the dependent chain of
// the select kernel (eight of them per launch), so they are compiled in only with -DDA_PHASE_TIMERS (make PHASE_TIMERS=1).
#ifdef DA_PHASE_TIMERS
#define UPD_TIMER_DECL long long up[5] = {0, 0, 0, 0, 0}, u0 = clock64(), u1;
#define UPD_TIMER_MARK(i) \
    u1 = clock64();       \
    up[i] += u1 - u0;     \
    u0 = u1;
#ifndef DA_TIMER_STEP_LO
#define DA_TIMER_STEP_LO 0  // the phase timers accumulate over the steps [LO, HI) of a chain (a window of the chain: -DDA_TIMER_STEP_LO=.. -DDA_TIMER_STEP_HI=..)
#endif
#ifndef DA_TIMER_STEP_HI
#define DA_TIMER_STEP_HI 0x7FFFFFFF
#endif
#define DA_TIMED_STEP(t) ((t) >= DA_TIMER_STEP_LO && (t) < DA_TIMER_STEP_HI)
#define UPD_TIMER_FLUSH \
    if (lane == 0 && DA_TIMED_STEP(g->iter - 1))      \
        for (int q = 0; q < 5; ++q) atomicAdd(&g->st_phase[7 + q], (unsigned long long)up[q]);
#define SEL_TIMER_DECL long long tp[8];
#define SEL_TIMER_MARK(i) tp[i] = clock64();
#define SEL_TIMER_FLUSH \
    if (DA_TIMED_STEP(iter)) for (int q = 0; q < 7; ++q) g->st_phase[q] += (unsigned long long)(tp[q + 1] - tp[q]);
#define UPD_TIMER_PASS_END UPD_TIMER_MARK(4)
#elif defined(DA_UPD_TAIL)
// -DDA_UPD_TAIL (diagnostic build, tools/gpu_tail.py): how long the passes of k_iter_update's wavefronts last, by kind -- st_phase[7] cycles of all
// passes, [8] passes, [9] cycles / [10] number of the passes with a rare case (a key beyond its first bucket, a creation in a full bucket), [11] passes
// with a block creation and no rare case, st_qdiag[0] their cycles, [1] / [4] passes longer than 20 k / 40 k cycles, [5] lanes of groups with a rare case
#ifndef DA_TIMER_STEP_LO
#define DA_TIMER_STEP_LO 0
#endif
#ifndef DA_TIMER_STEP_HI
#define DA_TIMER_STEP_HI 0x7FFFFFFF
#endif
#define DA_TIMED_STEP(t) ((t) >= DA_TIMER_STEP_LO && (t) < DA_TIMER_STEP_HI)
#define UPD_TIMER_DECL long long up[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, u0 = clock64(), u1;
#define UPD_TIMER_MARK(i)
#define UPD_TIMER_PASS_END                                                        \
    {                                                                             \
        u1 = clock64();                                                           \
        const long long dt = u1 - u0;                                             \
        const bool cr = __ballot(gnew) != 0;                                      \
        up[0] += dt, up[1] += 1;                                                  \
        if (rare) up[2] += dt, up[3] += 1, up[8] += __popcll(rare);               \
        if (!rare && cr) up[4] += 1, up[5] += dt;                                 \
        up[6] += dt > 20000, up[7] += dt > 40000;                                 \
        u0 = clock64();                                                           \
    }
#define UPD_TIMER_FLUSH                                                           \
    if (lane == 0 && DA_TIMED_STEP(g->iter - 1)) {                                \
        for (int q = 0; q < 5; ++q) atomicAdd(&g->st_phase[7 + q], (unsigned long long)up[q]); \
        atomicAdd(&g->st_qdiag[0], (unsigned long long)up[5]);                    \
        atomicAdd(&g->st_qdiag[1], (unsigned long long)up[6]);                    \
        atomicAdd(&g->st_qdiag[4], (unsigned long long)up[7]);                    \
        atomicAdd(&g->st_qdiag[5], (unsigned long long)up[8]);                    \
    }
#define SEL_TIMER_DECL
#define SEL_TIMER_MARK(i)
#define SEL_TIMER_FLUSH
#else
#define UPD_TIMER_DECL
#define UPD_TIMER_MARK(i)
#define UPD_TIMER_PASS_END
#define UPD_TIMER_FLUSH
#define SEL_TIMER_DECL
#define SEL_TIMER_MARK(i)
#define SEL_TIMER_FLUSH
#endif

// -DDA_STEP_CLOCKS (diagnostic build): where a chain's step goes -- the two kernels' durations as the chain sees them (first block start to last
// block end) and the two gaps between them (kernel boundary + the wait for a place on a CU), in ticks of the 100 MHz real-time counter, summed over
// the steps into st_qdiag[0] (select), [1] (select -> update), [4] (update), [5] (update -> next select), [7] (steps counted)
#ifdef DA_STEP_CLOCKS
#define CLK_NOW() ((unsigned long long)__builtin_amdgcn_s_memrealtime())
#define CLK_MARK(g, par, slot, inverted) atomicMax(&(g)->clk[par][slot], (inverted) ? ~CLK_NOW() : CLK_NOW())
#ifndef DA_CLK_STEP_LO
#define DA_CLK_STEP_LO 0  // the sums cover the steps [LO, HI) of every chain (a window of the chain)
#endif
#ifndef DA_CLK_STEP_HI
#define DA_CLK_STEP_HI 0x7FFFFFFF
#endif
#define CLK_TIMED_STEP(t) ((t) >= DA_CLK_STEP_LO && (t) < DA_CLK_STEP_HI)
#else
#define CLK_MARK(g, par, slot, inverted)
#endif

// header of a pair block's payload line (16 bytes, followed by the K u16 counts)
struct BlkHdr {
    int ov;         // n_overlap of the two rows (indexers.cc:36-56)
    float dl;       // |latency difference| of the two rows
    uint32_t rank;  // copy of hrank[slot]
    uint32_t idx;   // key index of the block's best key
};

// Pointers read from a chain descriptor are "generic" pointers to the compiler, which emits FLAT memory instructions
// for them.  FLAT operations count in BOTH vmcnt and lgkmcnt, so every LDS wait (lgkmcnt(0)) would also wait for all
// global loads in flight and serialise the software-pipelined loads of the greedy-loop kernels.  The hot kernels
// therefore keep their table / row / list pointers in the global address space (GLOBAL instructions, vmcnt only).
#define DA_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ T *gen(DA_GLOBAL T *p) { return (T *)p; }  // for the HIP atomic API
// Wave-uniform values pinned into scalar registers at this point of the program.  The greedy-loop kernels read ~30 fields
// of their chain descriptor; left alone, the compiler sinks each scalar load behind the early-exit branch that first needs
// it and the prologue becomes a chain of 4-6 dependent scalar round trips to L2 (ISA reading, round 2).  Loading every
// field first and pinning the values before the first branch makes it ONE round trip: an empty asm that "uses" the
// register is a point the load cannot be moved past.
// (the explicit v_readfirstlane moves a value that the compiler fetched with a vector load of a uniform address into a scalar
// register; on a value that already lives in one it folds away)
template <class T> __device__ __forceinline__ T da_uniform(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "pinned descriptor fields are 32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        const int r = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v));
        return __builtin_bit_cast(T, r);
    } else {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)u), hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(u >> 32));
        return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
    }
}
#define DA_PIN_ASM(x) do { x = da_uniform(x); asm volatile("" : "+s"(x)); } while (0)
template <class T> __device__ __forceinline__ void pin_one(T &v) { DA_PIN_ASM(v); }
template <class... T> __device__ __forceinline__ void pin_sgpr(T &...v) { (pin_one(v), ...); }
// The same for per-lane values, and a compiler-level fence for memory operations.  "Load early, use late": a loaded value
// that is first touched after pin_vgpr() / behind load_fence() is not waited for before that point, and loads in front of
// the fence are not sunk behind it -- the loads of a phase leave together instead of one round trip after the other.
#define DA_PIN_VASM(x) asm volatile("" : "+v"(x))
template <class T> __device__ __forceinline__ void pin_vone(T &v) { DA_PIN_VASM(v); }
template <class... T> __device__ __forceinline__ void pin_vgpr(T &...v) { (pin_vone(v), ...); }
__device__ __forceinline__ void load_fence() {
    asm volatile("" ::: "memory");         // the optimiser does not move memory operations across
    __builtin_amdgcn_sched_barrier(0);      // nor does the instruction scheduler move anything (e.g. a wait + v_readfirstlane of
}                                           // the first loaded value in front of the other loads)
__host__ __device__ __forceinline__ size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
typedef float da_f4 __attribute__((ext_vector_type(4)));
typedef int da_i4 __attribute__((ext_vector_type(4)));
typedef unsigned int da_u2 __attribute__((ext_vector_type(2)));
typedef unsigned long long da_ul2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ RowInfo load_row(const DA_GLOBAL RowInfo *rows, size_t i) {
    const da_f4 v = *reinterpret_cast<const DA_GLOBAL da_f4 *>(rows + i);
    return RowInfo{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ RowInfo pick_row(bool first, const RowInfo &a, const RowInfo &b) {  // field by field: the records stay in registers
    return RowInfo{first ? a.lo : b.lo, first ? a.hi : b.hi, first ? a.step : b.step, first ? a.lat : b.lat};
}
__device__ __forceinline__ void store_row(DA_GLOBAL RowInfo *rows, size_t i, const RowInfo &r) {
    *reinterpret_cast<DA_GLOBAL da_f4 *>(rows + i) = da_f4{r.lo, r.hi, r.step, r.lat};
}

// Row-list entry formats.  Narrow: u32 = col:8 | minus:12 | plus:12 (32 entries per 128-byte line); wide: 16 bytes.
template <class Cell> struct RowFmt;
template <> struct RowFmt<uint32_t> {
    using Entry = uint32_t;
    static __device__ __forceinline__ Entry pack(uint32_t col, uint32_t cell) { return (col << 24) | ((cell >> 16) << 12) | (cell & 0xFFFu); }
    static __device__ __forceinline__ uint32_t col(Entry e) { return e >> 24; }
    static __device__ __forceinline__ uint32_t cell(Entry e) { return (((e >> 12) & 0xFFFu) << 16) | (e & 0xFFFu); }
    static __device__ __forceinline__ Entry none() { return 0u; }
};
template <> struct RowFmt<uint64_t> {
    using Entry = da_ul2;  // x = cell, y = column
    static __device__ __forceinline__ Entry pack(uint32_t col, uint64_t cell) { return Entry{(unsigned long long)cell, (unsigned long long)col}; }
    static __device__ __forceinline__ uint32_t col(Entry e) { return (uint32_t)e.y; }
    static __device__ __forceinline__ uint64_t cell(Entry e) { return (uint64_t)e.x; }
    static __device__ __forceinline__ Entry none() { return Entry{0ull, 0ull}; }
};
// reference to a row as stored in the column lists and the partner list: row:24 | len:12 | off:28
constexpr int REF_ROW_BITS = 24, REF_LEN_BITS = 12, REF_OFF_BITS = 28;
__host__ __device__ __forceinline__ unsigned long long ref_pack(uint32_t row, uint32_t len, uint32_t off) {
    return (unsigned long long)row | ((unsigned long long)len << REF_ROW_BITS) | ((unsigned long long)off << (REF_ROW_BITS + REF_LEN_BITS));
}
__device__ __forceinline__ uint32_t ref_row(unsigned long long r) { return (uint32_t)r & ((1u << REF_ROW_BITS) - 1u); }
__device__ __forceinline__ uint32_t ref_len(unsigned long long r) { return (uint32_t)(r >> REF_ROW_BITS) & ((1u << REF_LEN_BITS) - 1u); }
__device__ __forceinline__ uint32_t ref_off(unsigned long long r) { return (uint32_t)(r >> (REF_ROW_BITS + REF_LEN_BITS)); }

// A table entry together with what a substitution needs of its two rows: the best entry a search found (64 bytes = one
// s_load_dwordx16 of the block that takes it up)
struct SpecPick {
    unsigned long long word;  // bound word of the entry (rank << 32 | tie >> 23); 0 = none / not usable
    unsigned long long tie;   // its full tie word (id1, id0, key index)
    da_u2 refA, refB;         // {offset, length} of the lists of rows id0 and id1
    RowInfo ra, rb;           // their records
};
static_assert(sizeof(SpecPick) == 64, "SpecPick is read as sixteen words");
struct CandEntry {  // a table entry by value: selection rank and tie word (id1, id0, key index)
    unsigned long long rank, tie;
};
#ifndef DA_CAND_CAP
#define DA_CAND_CAP 16  // measured (tools/spec_probe.cc, 64 / 128 square): never more than 8 of either.  (tests/emu builds with 2: its steps
#endif                  // overflow the lists often, so that the path on which the search block publishes the pick runs there all the time)
constexpr int CAND_CAP = DA_CAND_CAP, TOUCH_CAP = DA_CAND_CAP;

// Per-chain descriptor in device memory.  Pointers are raw device addresses into the arena.
// Per group of GS consecutive table slots (the search sweeps all of them every step):
struct alignas(32) GroupRec {
    unsigned long long ub;    // bound word (rank << 32 | tie_word >> 23) of the group's best entry: tight unless `glow` reaches it
    unsigned long long glow;  // the highest bound word a block's best entry had before an update LOWERED it, since the group was last verified: reaching the
                              // group's bound it says the bound is stale (group_note)
    unsigned long long gtie;  // full tie word of the group's best entry (valid while the group is clean)
    uint32_t dirty;           // 0: the tie word is exact (and the bound, unless glow reaches it); 1: an entry rose, the tie word is unknown
    uint32_t pad;
};
static_assert(sizeof(GroupRec) == 32, "group record");
struct ChainDev {
    // geometry (constant after set-up)
    int n_in, n_out, n_bits, K, Kpad, method, adder_size, carry_size;
    int rcap, lcap, gs_log2, n_groups;
    uint32_t C, cmask;
    // inputs
    const float *kernel;
    const float *qints;
    const float *lats;
    int32_t *xint;  // centred integer matrix [n_in][n_out]
    int8_t *shift0, *shift1;
    // state
    void *rlist;        // row-list entries (RowFmt<Cell>::Entry)
    da_u2 *rowoff;      // [rcap] {offset, length} of every row's list
    uint32_t rl_cap, rl_used;
    RowInfo *rows;
    uint32_t *stamp;
    unsigned long long *collist;
    int *collen;
    unsigned long long *hkey;
    uint32_t *hrank;
    unsigned char *hblk;  // [C] payload lines of (1 << pb_log2) bytes
    int pb_log2;
    GroupRec *grec;  // [n_groups] one 32-byte record per group of slots (two 16-byte loads per group in the search's sweep; an update's marks of a group land in ONE line)
    // per-iteration hand-off select -> update
    int *mcol;
    uint16_t *cmap;    // [n_out] 1 + index of a column among the substituted columns of this step, 0 = not substituted
    uint32_t *colbits; // [n_out][cb_words] bit r of column j: row r has digits in column j NOW.  The partner rows of a step are the
                       // OR of the substituted columns' bitmaps -- no list reads, no de-duplication, no stale entries
    int cb_words;
    uint32_t *pl_ids;  // [rcap] partner row ids of the step (before their list references are attached)
    void *mA, *mB;
    unsigned long long *plist;
    int m, n_partners;
    int claim_words;   // words of the optional LDS area in which the substitution block combines the row bitmaps of a young chain (0: none)
    uint32_t A, B, Nw;
    int pk_shift, pk_sub;  // the pick's key (shift, sub): the consumed B-role digits are the A-role ones moved by `shift`, signs flipped when `sub` (k_iter_update)
    // progress
    int n_rows, iter, done, error, unknown_hit;
    unsigned int n_live, n_used, live_peak;
    int4 *picks;
    // results of k_prepare
    int prep_nbits, prep_maxdcol;
    long long prep_digits, prep_pairs;
    // extraction
    uint32_t *fin_row;
    unsigned long long *fin_cell;
    uint32_t *fin_count;
    uint32_t *fin_start;           // [n_out + 1] exclusive prefix of fin_count
    uint32_t *pk_row;              // packed surviving digits, column major ...
    unsigned long long *pk_cell;   // ... and their cells
    float *pk_lat;                 // [n_rows] latency of every row
    int pk_cap;
    uint32_t pk_total;             // digits packed by k_pack (= fin_start[n_out])
    // column-sharded chains (cmvm_shard.h): the matrix handed to k_prepare has pn_out columns of which this chain holds
    // [col0, col0 + n_out); exchange buffers of the greedy step.  Ordinary chains: pn_out == n_out, col0 == 0.
    int pn_out, col0;
    int32_t *cs_init;             // [n_pairs][K] partial initial pair counts (summed over the ranks in place)
    int32_t *cs_flags;            // [(rcap + 3) / 4] 8-bit fields: row shares a substituted column
    uint32_t *cs_uni;             // [rcap] union of the partner rows of all ranks, ascending
    int cs_nuni;
    int32_t *cs_slab;             // [(3 cs_nuni + 6)][K] partial / summed count changes
    // statistics
    unsigned long long st_rescans, st_partners, st_matches, st_found, st_inserts, st_cells;
    // -log2f of non-power-of-two input steps (StepLog2, cmvm_core.h): table built by the host libm, nullptr / 0 when all steps are powers of two
    const uint32_t *step_mant;
    const float *step_tab;
    int n_step_mant;
    unsigned long long st_sel_bytes;  // algorithmic bytes of the selection steps (all but the group re-reads, which st_rescans prices)
    unsigned long long st_phase[12];  // shader-clock cycles per kernel phase (select: 0-6, update: 7-11)
    // ---- the next pick, known one step ahead (k_iter_select2: search_body / pick_body)
    SpecPick spec[2];              // [t & 1]: the best entry of the table of step t that touches neither row of step t's pick -- step t
                                   // does not change it --, left by the search block of step t for step t + 1 (word 0: none, or l_list overflowed)
    unsigned int c_n[2], l_n[2];   // [t & 1]: entries in c_list / l_list (c_n counts on past the capacity: an overflow is visible)
    CandEntry c_list[2][CAND_CAP]; // [t & 1]: every entry that touches a row of step t's pick or its new row and whose bound word reaches
                                   // spec[t & 1].word, as it stands AFTER step t: appended by the update of step t (zeroed by the selection of step t)
    CandEntry l_list[2][TOUCH_CAP];// [t & 1]: the entries of the table of step t that touch exactly one row of the pick and reach spec[t & 1].word,
                                   // listed by the search block: the update of step t passes on those it does not re-evaluate
    uint32_t *sp_cnt;              // [6][Kpad] exact counts of the six pairs among {A, B, new row}: select -> update
    unsigned long long st_fast;    // steps whose pick was known before the step began
    unsigned long long st_qphase[4];  // shader-clock cycles of the search block: bounds, arg-max (steps without a known pick), search, steps timed
#ifdef DA_STEP_CLOCKS
    unsigned long long clk[2][4];     // [step & 1]: ~(first select block start), last select block end, ~(first update block start), last update block end (s_memrealtime, 100 MHz)
#endif
    unsigned long long st_qdiag[9];   // (phase-timer builds) [5] steps whose work list held > 32 groups, [6] the longest work list, [7] passes over the whole table (pick not known ahead), [8] work-list entries in all; group re-reads that found a stale bound below the floor / of clean groups with an excluded
                                      // best entry; [2],[3] scratch; [4] sum over the steps of the longest wave's re-read rounds
};

__constant__ Log2Table c_log2;

// ------------------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x / WAVE; }

// Wave-wide reductions on the DPP path (row shifts inside the 16-lane rows, then the two row broadcasts of gfx9): six
// VALU operations per 32-bit word and no LDS round trip.  (The __shfl_xor butterflies they replace go through
// ds_bpermute -- about 900 cycles per 64-bit reduction, which dominated the arg-max loop of the selection.)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_u32(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
#define DA_DPP_REDUCE32(v, ident, OP)                                   \
    v = OP(v, dpp_u32<DPP_ROW_SHR1, 0xF>(ident, v));                    \
    v = OP(v, dpp_u32<DPP_ROW_SHR2, 0xF>(ident, v));                    \
    v = OP(v, dpp_u32<DPP_ROW_SHR4, 0xF>(ident, v));                    \
    v = OP(v, dpp_u32<DPP_ROW_SHR8, 0xF>(ident, v));                    \
    v = OP(v, dpp_u32<DPP_ROW_BCAST15, 0xA>(ident, v));                 \
    v = OP(v, dpp_u32<DPP_ROW_BCAST31, 0xC>(ident, v));
__device__ __forceinline__ uint32_t uadd32(uint32_t a, uint32_t b) { return a + b; }
// inclusive prefix sum over the wave (the same DPP sequence leaves the running sums in every lane)
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
    DA_DPP_REDUCE32(v, 0u, uadd32)
    return v;
}
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t smin32(uint32_t a, uint32_t b) { return (int)a < (int)b ? a : b; }
__device__ __forceinline__ int wave_min_i32(int v) {
    uint32_t x = (uint32_t)v;
    DA_DPP_REDUCE32(x, 0x7FFFFFFFu, smin32)
    return __builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    DA_DPP_REDUCE32(v, 0u, umax32)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v) {
    const uint32_t lo = dpp_u32<CTRL, ROW_MASK>(0u, (uint32_t)v), hi = dpp_u32<CTRL, ROW_MASK>(0u, (uint32_t)(v >> 32));
    const unsigned long long t = ((unsigned long long)hi << 32) | lo;
    return t > v ? t : v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = dpp_max_u64<DPP_ROW_SHR1, 0xF>(v);
    v = dpp_max_u64<DPP_ROW_SHR2, 0xF>(v);
    v = dpp_max_u64<DPP_ROW_SHR4, 0xF>(v);
    v = dpp_max_u64<DPP_ROW_SHR8, 0xF>(v);
    v = dpp_max_u64<DPP_ROW_BCAST15, 0xA>(v);
    v = dpp_max_u64<DPP_ROW_BCAST31, 0xC>(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
// Make this wave's LDS traffic visible to its own lanes: wait for outstanding LDS operations only (lgkmcnt(0));
// global loads stay in flight.  LDS operations of one wave complete in order, so no wider fence is needed for the
// per-wave counters; the wave barriers stop the compiler from moving LDS accesses across.
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}

// Slot hash of a row pair.  32-bit arithmetic only: since the partner rows moved from wavefronts to 16-lane groups the
// hash is computed in vector registers (twice per group and pass), and the 64-bit multiplies of a 64-bit finaliser were
// ~15 % of the update kernel's vector instructions.  Two odd multipliers combine the ids, then the "lowbias32" finaliser;
// the placement of a block never influences a result, only probe lengths.
__device__ __forceinline__ uint32_t hash_pair(uint32_t lo, uint32_t hi) {
    uint32_t h = lo * 0x9E3779B1u + hi * 0x85EBCA77u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned long long pack_pair(uint32_t lo, uint32_t hi) {
    return ((unsigned long long)hi << 32) | lo;
}
// group bound word: selection rank in the high half, the 32 most significant bits of the tie word below it, so
// that among equal ranks only groups holding the newest rows have to be inspected
__device__ __forceinline__ unsigned long long bound_word(uint32_t rank, unsigned long long tie) {
    return ((unsigned long long)rank << 32) | (tie >> 23);
}

// ------------------------------------------------------------------------------------------------ table ops
// Register-resident copy of the constant part of a chain descriptor (wave-uniform, lives in SGPRs); `g` is used
// for the few mutable counters only.
struct Ctx {
    int n_out, n_bits, K, Kpad, method, gs_log2, pb_log2;
    uint32_t cmask, windows;
    DA_GLOBAL unsigned long long *hkey;
    DA_GLOBAL uint32_t *hrank;
    DA_GLOBAL unsigned char *hblk;
    DA_GLOBAL GroupRec *grec;
    const DA_GLOBAL RowInfo *rows;
    ChainDev *g;  // derived from the kernel argument: already known to be global
    unsigned long long tomb;  // this launch's tombstone value
    unsigned long long rword;  // k_iter_update: bound word of the best entry this step leaves untouched (0: none); the best entry of every block
    unsigned int *cn;          // written or re-evaluated whose bound word reaches it is appended to cl[(*cn)++] (fold_entry): together with that
    CandEntry *cl;             // entry they are all the next selection has to compare
};

// launch_id: 2 * iteration for k_iter_select2, 2 * iteration + 1 for k_iter_update (only the low two bits are used)
__device__ __forceinline__ Ctx make_ctx_raw(ChainDev *g, int launch_id) {
    Ctx c;
    c.tomb = KEY_TOMB - (unsigned long long)(launch_id & 3);
    c.n_out = g->n_out;
    c.n_bits = g->n_bits;
    c.K = g->K;
    c.Kpad = g->Kpad;
    c.method = g->method;
    c.gs_log2 = g->gs_log2;
    c.pb_log2 = g->pb_log2;
    c.cmask = g->cmask;
    c.windows = g->C;  // raw here (nothing is computed between the descriptor loads: they leave as ONE batch); ctx_finish() scales it
    c.hkey = (DA_GLOBAL unsigned long long *)g->hkey;
    c.hrank = (DA_GLOBAL uint32_t *)g->hrank;
    c.hblk = (DA_GLOBAL unsigned char *)g->hblk;
    c.grec = (DA_GLOBAL GroupRec *)g->grec;
    c.rows = (const DA_GLOBAL RowInfo *)g->rows;
    c.g = g;
    c.rword = 0;
    c.cn = nullptr;
    c.cl = nullptr;
    return c;
}
__device__ __forceinline__ void ctx_finish(Ctx &c) { c.windows = c.windows / WAVE ? c.windows / WAVE : 1; }  // after the fields are pinned
__device__ __forceinline__ Ctx make_ctx(ChainDev *g, int launch_id) {
    Ctx c = make_ctx_raw(g, launch_id);
    ctx_finish(c);
    return c;
}
// payload line of a slot: 16-byte header, then the counts
__device__ __forceinline__ DA_GLOBAL unsigned char *blk_ptr(const Ctx &c, int slot) { return c.hblk + ((size_t)(uint32_t)slot << c.pb_log2); }
__device__ __forceinline__ BlkHdr load_hdr(const Ctx &c, int slot) {
    const da_i4 v = *reinterpret_cast<const DA_GLOBAL da_i4 *>(blk_ptr(c, slot));
    return BlkHdr{v.x, __int_as_float(v.y), (uint32_t)v.z, (uint32_t)v.w};
}
__device__ __forceinline__ void store_hdr(const Ctx &c, int slot, int ov, float dl, uint32_t rank, uint32_t idx) {
    *reinterpret_cast<DA_GLOBAL da_i4 *>(blk_ptr(c, slot)) = da_i4{ov, __float_as_int(dl), (int)rank, (int)idx};
}
__device__ __forceinline__ void store_best(const Ctx &c, int slot, uint32_t rank, uint32_t idx) {  // rank copy + best key index
    *reinterpret_cast<DA_GLOBAL da_u2 *>(blk_ptr(c, slot) + 8) = da_u2{rank, idx};
}
__device__ __forceinline__ uint32_t load_best_idx(const Ctx &c, uint32_t slot) { return *reinterpret_cast<const DA_GLOBAL uint32_t *>(blk_ptr(c, (int)slot) + 12); }
__device__ __forceinline__ DA_GLOBAL uint16_t *blk_cnt(const Ctx &c, int slot) { return reinterpret_cast<DA_GLOBAL uint16_t *>(blk_ptr(c, slot) + 16); }

// All table operations are executed by one full wavefront.  The table is probed in BUCKETS of 16 slots (one 128-byte
// line of keys), starting at the bucket the hash points into and continuing with the following buckets.  A key is
// always stored in the first bucket of its sequence that had a free slot, and slots never return to EMPTY, so a
// look-up stops at the first bucket that contains the key or an EMPTY slot.
constexpr uint32_t BUCKET = 16;

// look-up continuing at bucket number `first_bucket` of the sequence, 4 buckets (64 lanes) per step; slot or -1
__device__ int table_find_from(const Ctx &c, unsigned long long key, uint32_t h, uint32_t first_bucket) {
    const int lane = lane_id();
    const uint32_t b0 = (h & ~(BUCKET - 1)) + first_bucket * BUCKET;
    for (uint32_t w = 0; w < c.windows + 1; ++w) {
        uint32_t s = (b0 + w * WAVE + lane) & c.cmask;
        unsigned long long kk = c.hkey[s];
        unsigned long long hit = __ballot(kk == key);
        if (hit) return (int)((b0 + w * WAVE + (__ffsll((long long)hit) - 1)) & c.cmask);
        if (__ballot(kk == KEY_EMPTY)) return -1;
    }
    return -1;
}
__device__ __forceinline__ int table_find(const Ctx &c, unsigned long long key, uint32_t h) { return table_find_from(c, key, h, 0); }
// claim a slot for a key that is known to be absent: the first free (EMPTY or TOMB) slot in bucket order; -1 = full
__device__ int table_claim(const Ctx &c, unsigned long long key, uint32_t h) {
    const int lane = lane_id();
    const uint32_t b0 = h & ~(BUCKET - 1);
    for (uint32_t w = 0; w < c.windows + 1; ++w) {
        uint32_t s = (b0 + w * WAVE + lane) & c.cmask;
        unsigned long long kk = c.hkey[s];
        unsigned long long avail = __ballot(kk == KEY_EMPTY || (kk >= KEY_TOMB_LO && kk != c.tomb));
        while (avail) {
            int l = __ffsll((long long)avail) - 1;
            avail &= avail - 1;
            int ok = 0;
            if (lane == l) {
                ok = atomicCAS(gen(&c.hkey[s]), kk, key) == kk;
            }
            ok = __shfl(ok, l);
            if (ok) return (int)((b0 + w * WAVE + l) & c.cmask);
        }
    }
    return -1;
}

// A block's best (rank, key) changed from bound word `w_old` to `w_new` (0 = not selectable).  Invariant kept: a group's bound is TIGHT (the
// bound word of its best entry) unless glow[grp] reaches it; flags 0 = the stored tie word is exact too.
//   * the entry rose: it may be the group's best now -- raise the bound, tie word unknown (flag);
//   * it fell: remember the highest value that fell (glow).  If that reaches the group's bound, the group's best may have been lowered and the
//     bound is stale -- the selection sees it when it reads the bounds and re-reads such groups early, in wavefronts that would idle otherwise.
//     (Left to be found lazily, stale bounds pile up below the selection's floor and are all met together when the floor drops a rank class:
//     work lists of up to 4000 groups, 1 % of the launches > 140 us, measured in round 5.  Deciding it HERE needs the group's bound in the update
//     kernel: two more loads and four more registers per partner pass, + 1.5 .. 4 us per launch, measured.)
// Fire-and-forget atomics, nothing is read.
__device__ __forceinline__ void group_note(const Ctx &c, int slot, unsigned long long w_old, unsigned long long w_new) {
    const int grp = slot >> c.gs_log2;
    if (w_new > w_old) {
        atomicMax(gen(&c.grec[grp].ub), w_new);
#ifdef DA_AB_PLAIN_FLAG
        c.grec[grp].dirty = 1u;
#else
        atomicOr(gen(&c.grec[grp].dirty), 1u);
#endif
    } else {
#ifdef DA_AB_NO_GLOW  // (A/B: the lazy scheme -- still exact, stale bounds are found when the floor meets them)
        c.grec[grp].dirty = 1u;
#else
        atomicMax(gen(&c.grec[grp].glow), w_old);
#endif
    }
}
// one lane: the best entry (rank > 0) of a block as it stands after this launch -- a candidate for the next pick if it reaches the
// entry the step leaves untouched (rare by construction: at most a handful per step)
__device__ __forceinline__ void fold_entry(const Ctx &c, uint32_t rank, unsigned long long tie) {
    if (c.rword && bound_word(rank, tie) >= c.rword) {
        const unsigned int at = atomicAdd(c.cn, 1u);
        if (at < (unsigned int)CAND_CAP) {
            c.cl[at].rank = rank;
            c.cl[at].tie = tie;
        }
    }
}
// best-key index of every slot as a dense byte array behind hrank (the selection's search reads ranks, keys and indices of a whole
// group in one round trip; the copy in the payload header is what the update kernel reads with the counts)
__device__ __forceinline__ DA_GLOBAL uint8_t *hidx_ptr(const Ctx &c) { return reinterpret_cast<DA_GLOBAL uint8_t *>(c.hrank + ((size_t)c.cmask + 1)); }

// Store a complete block.  cnt_of(k) gives the count of key k; returns false when the table is full.
// Precondition: at least one count >= 2 (checked by the caller), key absent.  ra / rb: intervals of rows lo / hi.
// `w_out` (optional): the bound word of the new block's best entry, 0 when none is selectable (wave-uniform).
// `tally`: count the new block in the chain's n_live here, with a device atomic.  k_iter_update passes false and adds its creations and deletions
// ONCE PER WORKGROUP: in the first thousands of steps of a chain hundreds of blocks are created and deleted per step, and one atomic each on the
// same line of the chain descriptor serialises in the L2 at ~12 ns apiece -- behind which every later load of the issuing wave waits (vmcnt is
// in-order): measured in round 6, k_iter_update of steps 0 - 2000 took 24.8 us for one chain where the late steps take 6.7.
template <class CntFn>
__device__ bool table_insert(const Ctx &c, uint32_t lo, uint32_t hi, const RowInfo &ra, const RowInfo &rb, CntFn cnt_of, unsigned long long *w_out = nullptr, bool tally = true) {
    int lane = lane_id();
    unsigned long long key = pack_pair(lo, hi);
    int slot = table_claim(c, key, hash_pair(lo, hi));
    if (slot < 0) {
        if (lane == 0) c.g->error = E_TABLE_CAPACITY;
        return false;
    }
    int ov = n_overlap(ra, rb);
    float dl = fabsf(ra.lat - rb.lat);
    unsigned long long best = 0;
    DA_GLOBAL uint16_t *cnt = blk_cnt(c, slot);
    for (int k = lane; k < c.Kpad; k += WAVE) {
        uint32_t n = k < c.K ? cnt_of(k) : 0u;
        if (n > 65535u) c.g->error = E_COUNT_OVERFLOW;
        cnt[k] = (uint16_t)n;
        uint32_t r = entry_rank(n, ov, dl, c.method);
        unsigned long long cand = r ? (((unsigned long long)r << 8) | (unsigned)k) : 0ull;
        best = cand > best ? cand : best;
    }
    best = wave_max_u64(best);
    if (w_out) *w_out = (best >> 8) ? bound_word((uint32_t)(best >> 8), tie_word(lo, hi, (int)(best & 0xFF))) : 0ull;
    if (lane == 0) {
        uint32_t rank = (uint32_t)(best >> 8), idx = (uint32_t)(best & 0xFF);
        store_hdr(c, slot, ov, dl, rank, idx);
        c.hrank[slot] = rank;
        hidx_ptr(c)[slot] = (uint8_t)idx;
        if (rank) {
            group_note(c, slot, 0ull, bound_word(rank, tie_word(lo, hi, (int)idx)));
            fold_entry(c, rank, tie_word(lo, hi, (int)idx));
        }
        if (tally) atomicAdd(&c.g->n_live, 1u);  // no return value: fire-and-forget (the peak is sampled by the selection)
    }
    return true;
}

// lane 0: publish the re-evaluated best key of a block (or delete the block when no count >= 2 is left)
// bound word of a block's best entry as its header holds it
__device__ __forceinline__ unsigned long long hdr_word(unsigned long long key, uint32_t rank, uint32_t idx) {
    return rank ? bound_word(rank, tie_word((uint32_t)key, (uint32_t)(key >> 32), (int)idx)) : 0ull;
}
__device__ __forceinline__ void block_commit(const Ctx &c, int slot, unsigned long long key, const BlkHdr &h, unsigned long long best, int alive, bool tally = true) {
    const unsigned long long w_old = hdr_word(key, h.rank, h.idx);
    if (!alive) {
        c.hrank[slot] = 0;
        c.hkey[slot] = c.tomb;
        if (tally) atomicSub(&c.g->n_live, 1u);  // (see table_insert)
        if (h.rank) group_note(c, slot, w_old, 0ull);
        return;
    }
    const uint32_t rank = (uint32_t)(best >> 8), idx = (uint32_t)(best & 0xFF);
    if (rank != h.rank) c.hrank[slot] = rank;
    if (rank != h.rank || idx != h.idx) store_best(c, slot, rank, idx);
    if (idx != h.idx) hidx_ptr(c)[slot] = (uint8_t)idx;
    if (rank != h.rank || (rank && idx != h.idx))
        group_note(c, slot, w_old, rank ? bound_word(rank, tie_word((uint32_t)key, (uint32_t)(key >> 32), (int)idx)) : 0ull);
    if (rank) fold_entry(c, rank, tie_word((uint32_t)key, (uint32_t)(key >> 32), (int)idx));  // changed or not: the block touches a row of the step
}

// Re-evaluate a block after its counts changed (new_cnt(k, old) -> new count); deletes it when no count >= 2.
// Returns the bound word of the block's best entry afterwards (0: deleted, or nothing selectable); wave-uniform.
template <class CntFn>
__device__ unsigned long long table_update(const Ctx &c, int slot, unsigned long long key, CntFn new_cnt, bool tally = true, int *deleted = nullptr) {
    int lane = lane_id();
    const BlkHdr h = load_hdr(c, slot);
    DA_GLOBAL uint16_t *cnt = blk_cnt(c, slot);
    unsigned long long best = 0;
    int alive = 0;
    for (int k = lane; k < c.K; k += WAVE) {
        uint32_t old = cnt[k];
        uint32_t n = new_cnt(k, old);
        if (n != old) cnt[k] = (uint16_t)n;
        alive |= n >= 2;
        uint32_t r = entry_rank(n, h.ov, h.dl, c.method);
        unsigned long long cand = r ? (((unsigned long long)r << 8) | (unsigned)k) : 0ull;
        best = cand > best ? cand : best;
    }
    best = wave_max_u64(best);
    alive = __any(alive);
    if (lane == 0) block_commit(c, slot, key, h, best, alive, tally);
    if (deleted) *deleted = !alive;
    return alive && (best >> 8) ? bound_word((uint32_t)(best >> 8), tie_word((uint32_t)key, (uint32_t)(key >> 32), (int)(best & 0xFF))) : 0ull;
}

// ------------------------------------------------------------------------------------------------ k_prepare
// One block per chain: centring shifts, centred integer matrix, digit width, digit statistics.
__global__ void __launch_bounds__(256) k_prepare(ChainDev *chains) {
    ChainDev &ch = chains[blockIdx.x];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_dcol = reinterpret_cast<int *>(smem);  // [n_out]
    __shared__ unsigned int s_max;
    __shared__ unsigned long long s_digits, s_pairs;
    __shared__ int s_maxd;
    const int n_in = ch.n_in, n_out = ch.pn_out;  // the whole matrix (a column-sharded chain holds a slice of it)
    const float *k = ch.kernel;
    if (threadIdx.x == 0) {
        s_max = 0;
        s_digits = 0;
        s_pairs = 0;
        s_maxd = 0;
    }
    for (int j = threadIdx.x; j < n_out; j += blockDim.x) {
        int low = 127;
        for (int i = 0; i < n_in; ++i) low = min(low, lsb_loc(k[(size_t)i * n_out + j]));
        ch.shift1[j] = (int8_t)low;
        s_dcol[j] = 0;
    }
    __syncthreads();
    int nw = blockDim.x / WAVE;
    for (int i = wave_id(); i < n_in; i += nw) {
        int low = 127;
        for (int j = lane_id(); j < n_out; j += WAVE) low = min(low, lsb_loc(ldexpf(k[(size_t)i * n_out + j], -(int)ch.shift1[j])));
        low = wave_min_i32(low);
        if (lane_id() == 0) ch.shift0[i] = (int8_t)low;
        bool dead = ch.qints[3 * i] == 0.0f && ch.qints[3 * i + 1] == 0.0f;
        unsigned int mx = 0;
        for (int j = lane_id(); j < n_out; j += WAVE) {
            float v = ldexpf(ldexpf(k[(size_t)i * n_out + j], -(int)ch.shift1[j]), -low);
            int32_t x = (int32_t)v;
            ch.xint[(size_t)i * n_out + j] = x;
            mx = max(mx, (unsigned int)(x < 0 ? -x : x));
            if (!dead && x != 0) atomicAdd(&s_dcol[j], naf_weight(x));
        }
        mx = wave_max_u32(mx);
        if (lane_id() == 0) atomicMax(&s_max, mx);
    }
    __syncthreads();
    unsigned long long dig = 0, pairs = 0;
    int maxd = 0;
    for (int j = threadIdx.x; j < n_out; j += blockDim.x) {
        int d = s_dcol[j];
        dig += d;
        pairs += (unsigned long long)d * (d > 0 ? d - 1 : 0) / 2;
        maxd = max(maxd, d);
    }
    atomicAdd(&s_digits, dig);
    atomicAdd(&s_pairs, pairs);
    atomicMax(&s_maxd, maxd);
    __syncthreads();
    if (threadIdx.x == 0) {
        ch.prep_nbits = csd_width(s_max);
        ch.prep_digits = (long long)s_digits;
        ch.prep_pairs = (long long)s_pairs;
        ch.prep_maxdcol = s_maxd;
    }
}

// ------------------------------------------------------------------------------------------------ k_init_cells
// grid (ceil(n_out / 4), n_chains): one wave per column builds the entries of that column in the (dense) lists of the
// input rows and the column's row list.
// grid (blocks, chains): the state every chain starts from -- claim stamps, ranks, group bounds and row bitmaps zero, keys
// EMPTY (all ones), every group dirty.  16-byte stores; every array starts on a 256-byte boundary of the arena and is followed
// by padding up to the next one (Carver), so the last, partial 16 bytes of an array may be written whole.
__device__ __forceinline__ void fill16(void *p, size_t bytes, uint32_t v, size_t t0, size_t stride) {
    da_i4 *q = reinterpret_cast<da_i4 *>(p);
    const da_i4 w = da_i4{(int)v, (int)v, (int)v, (int)v};
    for (size_t i = t0, n = (bytes + 15) / 16; i < n; i += stride) q[i] = w;
}
__global__ void __launch_bounds__(256) k_init_state(ChainDev *chains) {
    const ChainDev &ch = chains[blockIdx.y];
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    fill16(ch.stamp, sizeof(uint32_t) * (size_t)ch.rcap, 0u, t0, stride);
    fill16(ch.hkey, sizeof(unsigned long long) * (size_t)ch.C, 0xFFFFFFFFu, t0, stride);
    fill16(ch.hrank, sizeof(uint32_t) * (size_t)ch.C, 0u, t0, stride);
    for (size_t i = t0; i < (size_t)ch.n_groups; i += stride) ch.grec[i] = GroupRec{0ull, 0ull, 0ull, 1u, 0u};
    fill16(ch.colbits, sizeof(uint32_t) * (size_t)ch.n_out * ch.cb_words, 0u, t0, stride);
}

template <class Cell> __global__ void __launch_bounds__(256) k_init_cells(ChainDev *chains) {
    using O = CellOps<Cell>;
    using F = RowFmt<Cell>;
    ChainDev &ch = chains[blockIdx.y];
    int j = blockIdx.x * (blockDim.x / WAVE) + wave_id();
    if (j >= ch.n_out) return;
    auto *rl = reinterpret_cast<typename F::Entry *>(ch.rlist);
    int lane = lane_id(), len = 0;
    for (int base = 0; base < ch.n_in; base += WAVE) {
        int i = base + lane;
        Cell c = 0;
        if (i < ch.n_in) {
            bool dead = ch.qints[3 * i] == 0.0f && ch.qints[3 * i + 1] == 0.0f;
            uint32_t p, m;
            naf_masks(ch.xint[(size_t)i * ch.pn_out + ch.col0 + j], p, m);
            c = dead ? (Cell)0 : O::make(p, m);
            rl[(size_t)i * ch.n_out + j] = F::pack((uint32_t)j, c);  // dense row: entry j is column j, empty cells included
        }
        unsigned long long nz = __ballot(c != 0);
        if (lane < 2 && base + 32 * lane < ch.n_in) ch.colbits[(size_t)j * ch.cb_words + (base >> 5) + lane] = (uint32_t)(nz >> (32 * lane));
        if (c != 0)
            ch.collist[(size_t)j * ch.lcap + len + __popcll(nz & ((1ull << lane) - 1))] = ref_pack((uint32_t)i, (uint32_t)ch.n_out, (uint32_t)i * (uint32_t)ch.n_out);
        len += __popcll(nz);
    }
    if (lane == 0) ch.collen[j] = len;
    if (j == 0)
        for (int i = lane; i < ch.n_in; i += WAVE) {
            ch.rows[i] = RowInfo{ch.qints[3 * i], ch.qints[3 * i + 1], ch.qints[3 * i + 2], ch.lats[i]};
            ch.rowoff[i] = da_u2{(uint32_t)i * (uint32_t)ch.n_out, (uint32_t)ch.n_out};
        }
}

// exact pair counts of one pair of INPUT rows (dense lists) over all columns into LDS counters (one wave)
template <class Cell>
__device__ __forceinline__ void count_row_pair(const Ctx &c, const typename RowFmt<Cell>::Entry *rl, uint32_t lo, uint32_t hi, uint32_t *cnt) {
    using F = RowFmt<Cell>;
    int lane = lane_id();
    for (int k = lane; k < c.Kpad; k += WAVE) cnt[k] = 0;
    lds_fence();
    const auto *el = rl + (size_t)lo * c.n_out, *eh = rl + (size_t)hi * c.n_out;
    for (int j = lane; j < c.n_out; j += WAVE) {
        Cell a = F::cell(el[j]);
        if (!a) continue;
        if (lo == hi)
            for_pairs_self<Cell>(a, c.n_bits, [&](int k) { atomicAdd(&cnt[k], 1u); });
        else {
            Cell b = F::cell(eh[j]);
            if (b) for_pairs_cross<Cell>(a, b, c.n_bits, [&](int k) { atomicAdd(&cnt[k], 1u); });
        }
    }
    lds_fence();
}
__device__ __forceinline__ bool wave_any_ge2(const uint32_t *cnt, int K) {
    int f = 0;
    for (int k = lane_id(); k < K; k += WAVE) f |= cnt[k] >= 2;
    return __any(f);
}

// ------------------------------------------------------------------------------------------------ k_init_pairs
// grid (ceil(n_pairs / 4), n_chains): one wave per initial row pair (i0 <= i1).
template <class Cell> __global__ void __launch_bounds__(256) k_init_pairs(ChainDev *chains) {
    ChainDev *g = &chains[blockIdx.y];
    if (g->method == M_DUMMY) return;
    const Ctx c = make_ctx(g, 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem) + (size_t)wave_id() * c.Kpad;
    long long n_in = g->n_in;
    long long n_pairs = n_in * (n_in + 1) / 2;
    long long p = (long long)blockIdx.x * (blockDim.x / WAVE) + wave_id();
    if (p >= n_pairs) return;
    // p = i1 (i1 + 1) / 2 + i0
    long long i1 = (long long)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (i1 * (i1 + 1) / 2 > p) --i1;
    while ((i1 + 1) * (i1 + 2) / 2 <= p) ++i1;
    uint32_t hi = (uint32_t)i1, lo = (uint32_t)(p - i1 * (i1 + 1) / 2);
    const auto *rl = reinterpret_cast<const typename RowFmt<Cell>::Entry *>(g->rlist);
    count_row_pair<Cell>(c, rl, lo, hi, cnt);
    if (!wave_any_ge2(cnt, c.K)) return;
    if (c.method < 0) {  // unknown method string with a non-empty table: the reference throws here
        if (lane_id() == 0) g->unknown_hit = 1;
        return;
    }
    table_insert(c, lo, hi, load_row(c.rows, lo), load_row(c.rows, hi), [&](int k) { return cnt[k]; });
}

// ================================================================================= k_iter_select2: the next pick, one step ahead
// A greedy step changes only table entries that touch the two rows of its pick (A, B) or the row it creates (N).  So the
// best entry R_t of the table of step t that touches neither A_t nor B_t is still in the table of step t + 1, unchanged, and
// the pick of step t + 1 is the largest of R_t and the entries that touch A_t, B_t or N_t.  Of those only the ones that reach
// R_t matter, and there are hardly ever any (tools/spec_probe.cc: none in 88 % of the steps, never more than 8).  The selection
// of a step is split into two workgroups per chain that run side by side:
//   * the SEARCH block (search_body, blockIdx.y == 0) looks for R_t -- the lazy arg-max over the group bounds with the entries
//     of A_t / B_t left out -- and lists the entries that touch exactly one of A_t / B_t and reach R_t (l_list): spec[t & 1];
//   * the update of step t appends the best entry of every block it writes or re-evaluates (partner blocks, the six blocks among
//     {A, B, N}) that reaches R_t to c_list[t & 1], changed or not (fold_entry), and passes on the entries of l_list whose block
//     it does not visit (the other row is not a partner row: the entry is unchanged);
//   * the SUBSTITUTION block (pick_body, blockIdx.y == 1) of step t + 1 takes the largest of R_t and c_list[t & 1] as its pick:
//     no bounds, no arg-max in front of the substitution.  Only when there is no R_t or a list overflowed (the first step of a chain) does it
//     find the pick itself -- a plain arg-max over the dense rank array (table_argmax_block) --, as does the search block for its own purpose
//     (through the bounds): the two blocks never wait for each other.
// Exactness: an entry of the table of step t + 1 either touches none of A_t, B_t, N_t -- then it is at most R_t --, or its block
// was written / re-evaluated by the update -- then its best entry is in c_list if it reaches R_t --, or it touches A_t or B_t
// and its block was not visited -- then it is unchanged since the search saw it, and in l_list -> c_list if it reaches R_t.
// "Reaches" is decided on bound words (rank << 32 | tie >> 23), which order coarser than (rank, tie): a superset; the pick is
// the maximum by (rank, tie) of the listed entries and R_t.
// Only the search block reads or writes group bounds; the table itself is not written by this kernel at all (the six special
// blocks moved to k_iter_update), so the two blocks share nothing but the descriptor fields named above.
constexpr uint32_t ROW_NONE = 0xFFFFFFFFu;
#ifndef DA_SEL2_THREADS
#define DA_SEL2_THREADS 1024  // measured (MI355X, round 5): 512 threads are slower, alone and in the batch (fewer waves share the group reads)
#endif
constexpr int SEL2_THREADS = DA_SEL2_THREADS;  // threads of a k_iter_select2 block (both roles); MAX_GROUPS / SEL2_THREADS group bounds per lane of the search
static_assert(MAX_GROUPS % SEL2_THREADS == 0 && SEL2_THREADS % WAVE == 0 && SEL2_THREADS >= 256, "k_iter_select2 geometry");
#ifndef DA_AB_REPAIR_ROUNDS
#define DA_AB_REPAIR_ROUNDS 8  // measured (MI355X, C3 batch / one chain, us per step): 0: 36.0 / 24.9, 1: 35.5 / 24.0, 2: 34.9 / 23.6, 4: 34.4 / 23.1, 8: 34.4 / 22.9
#endif
#ifndef DA_SEL_RK
#define DA_SEL_RK 16  // (see DA_MAX_GROUPS)
#endif
constexpr int QL_CAP = DA_CAND_CAP * 4;  // entries touching the pick's rows the search may meet above its rising floor before it gives up (LDS)

#ifdef DA_PHASE_TIMERS
#define Q_TIMER_DECL long long qp[4];
#define Q_TIMER_MARK(i) qp[i] = clock64();
#else
#define Q_TIMER_DECL
#define Q_TIMER_MARK(i)
#endif

// The table's best entry -- highest rank, then highest tie word -- by ONE WORKGROUP and a plain pass over the dense rank array: no bounds, nothing
// written.  For the substitution block of a step whose pick is not known ahead (the first step of a chain; an overflowed candidate list): 4 bytes per
// slot twice (8 + 8 MB for a 256x256 chain, once per chain).  rank 0: nothing selectable.  red_*: one LDS word per wave.  Ends with a block barrier.
__device__ __forceinline__ void table_argmax_block(ChainDev *g, uint32_t &rank, unsigned long long &tie, uint32_t *red_rank, unsigned long long *red_tie) {
    const DA_GLOBAL uint32_t *hrank = (const DA_GLOBAL uint32_t *)g->hrank;
    const DA_GLOBAL unsigned long long *hkey = (const DA_GLOBAL unsigned long long *)g->hkey;
    const uint32_t C = g->C;
    const DA_GLOBAL uint8_t *hidx = reinterpret_cast<const DA_GLOBAL uint8_t *>(hrank + (size_t)C);
    const int tid = threadIdx.x, nthr = blockDim.x, lane = lane_id(), wid = wave_id(), nw = nthr / WAVE;
    typedef unsigned int da_u4 __attribute__((ext_vector_type(4)));
    const DA_GLOBAL da_u4 *hr4 = reinterpret_cast<const DA_GLOBAL da_u4 *>(hrank);  // (C is a multiple of 256, the array starts on a 256-byte boundary)
    uint32_t best = 0;
    for (uint32_t i = (uint32_t)tid; i < C / 4; i += (uint32_t)nthr) {
        const da_u4 v = hr4[i];
        best = max(max(best, v.x), max(max(v.y, v.z), v.w));
    }
    best = wave_max_u32(best);
    if (lane == 0) red_rank[wid] = best;
    __syncthreads();
    best = wave_max_u32(lane < nw ? red_rank[lane] : 0u);
    unsigned long long bt = 0;
    if (best != 0)
        for (uint32_t i = (uint32_t)tid; i < C / 4; i += (uint32_t)nthr) {
            const da_u4 v = hr4[i];
            const uint32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r[q] == best) {
                    const uint32_t sl = 4 * i + (uint32_t)q;
                    const unsigned long long k = hkey[sl];
                    const unsigned long long tw = tie_word((uint32_t)k, (uint32_t)(k >> 32), (int)hidx[sl]);
                    bt = tw > bt ? tw : bt;
                }
        }
    bt = wave_max_u64(bt);
    if (lane == 0) red_tie[wid] = bt;
    __syncthreads();
    bt = wave_max_u64(lane < nw ? red_tie[lane] : 0ull);
    rank = best;
    tie = bt;
    __syncthreads();
}

// The pick of a step from what the previous step left: the best entry it did not touch (spec) against the candidates its update
// listed.  Every wave computes the same (one round trip when there are candidates, none otherwise).  false: no such entry, or a list
// overflowed -- the step finds its pick the long way.
__device__ __forceinline__ bool resolve_pick(const DA_GLOBAL CandEntry *cl, unsigned long long sp_word, unsigned long long sp_tie, unsigned int cn, unsigned long long &tie,
                                             bool &from_spec) {
    tie = sp_tie;
    from_spec = true;
    if (sp_word == 0 || cn > (unsigned int)CAND_CAP) return false;
    if (cn) {
        const int lane = lane_id();
        uint32_t r = 0;
        unsigned long long t = 0;
        if (lane < (int)cn) {
            r = (uint32_t)cl[lane].rank;
            t = cl[lane].tie;
        }
        const uint32_t br = wave_max_u32(r);
        const unsigned long long bt = wave_max_u64(r == br ? t : 0ull);
        const uint32_t rr = (uint32_t)(sp_word >> 32);
        if (br > rr || (br == rr && bt > sp_tie)) {
            tie = bt;
            from_spec = false;
        }
    }
    return true;
}

template <class Cell> __device__ __forceinline__ void search_body(ChainDev *g, int step) {
    constexpr int NW = SEL2_THREADS / WAVE, GPL = MAX_GROUPS / SEL2_THREADS;  // waves; group bounds per lane
    constexpr int RKN = DA_SEL_RK;  // slots per lane of a group read that are held in registers (x 64 lanes: groups of up to 512 slots in one round trip)
    const int par = step & 1;
    int was_done = g->done, had_error = g->error, n_groups = g->n_groups;
    unsigned long long sp_word = g->spec[par ^ 1].word, sp_tie = g->spec[par ^ 1].tie;
    unsigned int cn_prev = g->c_n[par ^ 1];
    Ctx c = make_ctx_raw(g, 2 * step);
    const DA_GLOBAL da_u2 *rowoff = (const DA_GLOBAL da_u2 *)g->rowoff;
    const DA_GLOBAL CandEntry *cl_prev = (const DA_GLOBAL CandEntry *)&g->c_list[par ^ 1][0];
    DA_GLOBAL CandEntry *ll_out = (DA_GLOBAL CandEntry *)&g->l_list[par][0];
    pin_sgpr(was_done, had_error, n_groups, sp_word, sp_tie, cn_prev);
    pin_sgpr(c.gs_log2, c.cmask, c.hkey, c.hrank, c.grec, c.rows);
    pin_sgpr(rowoff, cl_prev, ll_out);
    if (was_done || had_error != E_OK) return;  // (the substitution block stops the chain)
    __shared__ unsigned long long q_floor, q_red_tie[NW], q_ub[GPL][SEL2_THREADS];
    __shared__ uint32_t q_work[MAX_GROUPS];  // groups still to be read after round 0: index into q_ub | dirty << 31
    __shared__ uint32_t q_sl[MAX_GROUPS];    // groups whose bound is stale (their best entry was lowered): index into q_ub
    __shared__ unsigned int q_wn, q_sn, q_stake;
    __shared__ uint32_t q_red_rank[NW];
    __shared__ CandEntry q_L[QL_CAP];
    __shared__ unsigned int q_Ln, q_Lout;
    const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
    const int gs = 1 << c.gs_log2;
    const DA_GLOBAL uint8_t *hidx = hidx_ptr(c);
    Q_TIMER_DECL
    Q_TIMER_MARK(0)
    const int GPW = (n_groups + NW - 1) / NW;  // wave w owns the groups [w * GPW, (w + 1) * GPW)
    unsigned long long pk_tie;
    bool from_spec;
    const bool fast = resolve_pick(cl_prev, sp_word, sp_tie, cn_prev, pk_tie, from_spec);
    uint32_t exA = ROW_NONE, exB = ROW_NONE;  // rows whose entries the pass leaves out
    if (fast) {
        exA = (uint32_t)((pk_tie >> 7) & 0xFFFFFFu);
        exB = (uint32_t)(pk_tie >> 31);
    }
    unsigned int rescans = 0;
#ifdef DA_PHASE_TIMERS
    unsigned int q_stale = 0, q_touch = 0, q_rounds = 0;
#endif
    for (int pass = fast ? 1 : 0; pass < 2; ++pass) {
        // pass 0 (only when the pick is not known yet): the arg-max over the whole table -> the pick, published for the substitution block;
        // pass 1: the same with the entries of the pick's rows left out -> R; the entries that touch exactly one of the rows and reach the
        // floor of the moment are collected on the way (q_L)
        if (tid == 0) {
            q_floor = 0;
            q_Ln = 0;
            q_Lout = 0;
            q_wn = 0;
            q_sn = 0;
            q_stake = 0;
        }
        uint32_t nr = 0;  // this lane's best entry so far outside the excluded rows
        unsigned long long nt = 0;
        // returns the entry's bound word if it counts towards the floor, else 0; fl = the floor at the time (entries of the excluded rows are listed from there on)
        auto offer = [&](uint32_t r, unsigned long long tw, unsigned long long fl, bool list) -> unsigned long long {
            const uint32_t i0 = (uint32_t)((tw >> 7) & 0xFFFFFFu), i1 = (uint32_t)(tw >> 31);
            const bool t0 = i0 == exA || i0 == exB, t1 = i1 == exA || i1 == exB;
            if (!(t0 || t1)) {
                if (r > nr || (r == nr && tw > nt)) {
                    nr = r;
                    nt = tw;
                }
                return bound_word(r, tw);
            }
            if (list && t0 != t1 && bound_word(r, tw) >= fl) {
                const unsigned int at = atomicAdd(&q_Ln, 1u);
                if (at < (unsigned int)QL_CAP) q_L[at] = CandEntry{(unsigned long long)r, tw};
            }
            return 0ull;
        };
        // ---- ONE vector round trip: bound, dirty flag and stored tie word of this lane's (up to GPL) groups; unconditional loads, index clamped
        // (behind an `in ? load : 0` the compiler waits for each load inside its own branch).  A clean group's bound and tie word are exact: its best entry is a candidate right away, or -- if it touches an
        // excluded row -- the group has to be read.  (The second pass of a launch reads what the first one tightened: fenced below.)
        // Groups that may have to be read -- dirty, or clean with an excluded best entry -- keep their bound in LDS (q_ub[u][tid], 0 = nothing to
        // read); a lane holds only its highest one in registers (top, top_u) and one bit per group: dirty.
        unsigned long long cl = 0, top = 0;
        int top_u = 0;
        uint32_t dmask = 0, rmask = 0;  // per group of this lane: not verified (a read tightens it) / bound stale (read in any case)
        {
            unsigned long long ubv[GPL], gtr[GPL], glv[GPL];
            uint32_t dv[GPL];
#pragma unroll
            for (int u = 0; u < GPL; ++u) {
                const int q = wid * GPW + lane + u * WAVE;
                const int qc = (lane + u * WAVE < GPW && q < n_groups) ? q : 0;
                const DA_GLOBAL da_i4 *rp = reinterpret_cast<const DA_GLOBAL da_i4 *>(&c.grec[qc]);  // the record as two 16-byte loads
                const da_i4 r0 = rp[0], r1 = rp[1];
                ubv[u] = ((unsigned long long)(uint32_t)r0.y << 32) | (uint32_t)r0.x;
                glv[u] = ((unsigned long long)(uint32_t)r0.w << 32) | (uint32_t)r0.z;
                gtr[u] = ((unsigned long long)(uint32_t)r1.y << 32) | (uint32_t)r1.x;
                dv[u] = (uint32_t)r1.z;
            }
#pragma unroll
            for (int u = 0; u < GPL; ++u) {
                const int q = wid * GPW + lane + u * WAVE;
                const bool in = lane + u * WAVE < GPW && q < n_groups;
                unsigned long long cand = 0;  // bound of a group that may have to be read
                if (in && ubv[u] != 0 && glv[u] >= ubv[u]) {  // an entry that had reached the bound was lowered: the bound may be stale.  Read by a wave that has nothing else to do (or, like any
                    cand = ubv[u] ? ubv[u] : 1ull;  // group, when the bound reaches the floor): the backlog stays short and never arrives all at once
                    dmask |= 1u << u;
                    rmask |= 1u << u;
                } else if (in && ubv[u] != 0) {
                    if (dv[u]) {
                        cand = ubv[u];
                        dmask |= 1u << u;
                    } else if (ubv[u] >= cl) {  // (a clean group below this lane's best so far cannot matter, whatever rows its best entry touches)
                        const unsigned long long w = offer((uint32_t)(ubv[u] >> 32), gtr[u], 0ull, false);  // (an excluded best entry is listed when its group is read)
                        if (w)
                            cl = max(cl, w);
                        else
                            cand = ubv[u];
                    }
                }
                q_ub[u][tid] = cand;
                if (cand > top && !(rmask >> u & 1u)) {  // (round 0 takes the highest group that has to reach the floor; the stale ones go to the work list)
                    top = cand;
                    top_u = u;
                }
            }
        }
        if (pass == (fast ? 1 : 0)) { Q_TIMER_MARK(1) }
        __syncthreads();  // q_floor and the list counters zeroed
        cl = wave_max_u64(cl);
        if (lane == 0 && cl) atomicMax(&q_floor, cl);
#pragma unroll
        for (int u = 0; u < GPL; ++u)
            if (rmask >> u & 1u) q_sl[atomicAdd(&q_sn, 1u)] = (uint32_t)(u * SEL2_THREADS + tid);
        __syncthreads();
        // read one group (by one wavefront): every slot whose rank reaches the floor's is a candidate (or, touching an excluded row, listed); a
        // dirty group's bound is tightened.  (All ranks, keys and indices in ONE round trip was measured too: 114 registers per thread, and a
        // 1024-thread block of this kernel then needs a CU without a single k_iter_update block -- in the batch 1 % of the launches waited
        // > 200 us for one.  This form fits the 64 registers of eight waves per SIMD.)
        auto read_group = [&](const uint32_t grp, const bool dirty, const unsigned long long /*bound*/, const unsigned long long fl) {
            const uint32_t base = grp * gs;
            // ---- round trip 1: the ranks of all slots (one dense array, coalesced; 8 slots per lane)
            uint32_t rk[RKN], grank = 0;
#pragma unroll
            for (int u = 0; u < RKN; ++u) {
                const int o = lane + u * WAVE;
                rk[u] = c.hrank[base + (o < gs ? o : 0)];
            }
            load_fence();
#pragma unroll
            for (int u = 0; u < RKN; ++u) {
                pin_vgpr(rk[u]);
                if (lane + u * WAVE >= gs) rk[u] = 0;
                grank = max(grank, rk[u]);
            }
            for (int o = lane + RKN * WAVE; o < gs; o += WAVE) grank = max(grank, c.hrank[base + o]);
            grank = wave_max_u32(grank);
            // only slots whose rank reaches the floor's can matter (the floor's own rank included: the tie word decides)
            // (a group that is not verified is left exact: at least its best slots are read)
            const uint32_t fr = (uint32_t)(fl >> 32), thr0 = fr ? fr : 1u, thr = dirty && grank != 0 && grank < thr0 ? grank : thr0;
            unsigned long long gt = 0, lw = 0;
            if (grank >= thr) {  // (wave-uniform: a clean group whose excluded best entry was all it held that high costs nothing beyond the ranks)
                // ---- round trip 2: key and best-key index of those slots -- a lane takes its slots one per turn, all lanes' loads of a turn in
                // flight together (nearly always a single turn: a handful of the 512 slots qualify)
                uint32_t want = 0;
#pragma unroll
                for (int u = 0; u < RKN; ++u) want |= rk[u] >= thr ? 1u << u : 0u;
                while (__any(want != 0)) {
                    if (want) {
                        const int u = ctz32(want);
                        want &= want - 1;
                        uint32_t r = rk[0];
#pragma unroll
                        for (int v = 1; v < RKN; ++v) r = u == v ? rk[v] : r;
                        const uint32_t sl = base + (uint32_t)(lane + u * WAVE);
                        const unsigned long long k = c.hkey[sl];
                        const unsigned long long tw = tie_word((uint32_t)k, (uint32_t)(k >> 32), (int)hidx[sl]);
                        if (r == grank) gt = tw > gt ? tw : gt;
                        lw = max(lw, offer(r, tw, fl, pass != 0));
                    }
                }
                for (int o = lane + RKN * WAVE; o < gs; o += WAVE) {
                    const uint32_t r2 = c.hrank[base + o];
                    if (r2 >= thr) {
                        const unsigned long long k2 = c.hkey[base + o];
                        const unsigned long long tw = tie_word((uint32_t)k2, (uint32_t)(k2 >> 32), (int)hidx[base + o]);
                        if (r2 == grank) gt = tw > gt ? tw : gt;
                        lw = max(lw, offer(r2, tw, fl, pass != 0));
                    }
                }
                if (lw > fl) atomicMax(&q_floor, lw);
            }
            if (dirty) {  // the group's bound and tie word, exact again
                gt = wave_max_u64(gt);
                if (lane == 0) {
                    const unsigned long long nbw = grank ? bound_word(grank, gt) : 0ull;
                    DA_GLOBAL da_i4 *wp = reinterpret_cast<DA_GLOBAL da_i4 *>(&c.grec[grp]);  // {bound, glow = 0} | {tie word, flag = 0}: two 16-byte stores
                    wp[0] = da_i4{(int)(uint32_t)nbw, (int)(uint32_t)(nbw >> 32), 0, 0};
                    wp[1] = da_i4{(int)(uint32_t)gt, (int)(uint32_t)(gt >> 32), 0, 0};
                }
            }
            ++rescans;
#ifdef DA_PHASE_TIMERS
            q_stale += dirty && grank < thr0;
            q_touch += !dirty;
#endif
        };
        // ---- round 0: every wave reads its highest group that is dirty (bound possibly stale) or whose best entry is excluded, if that
        // bound reaches the floor of the clean groups.  The floor rises to (nearly) the answer.
        {
            const unsigned long long fl = __hip_atomic_load(&q_floor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long wtop = wave_max_u64(top);
            if (wtop != 0 && wtop >= fl) {
                const int owner = __ffsll((long long)__ballot(top == wtop)) - 1;
                const int own_u = __builtin_amdgcn_readlane(top_u, owner);
                const bool own_dirty = (((uint32_t)__builtin_amdgcn_readlane((int)dmask, owner) >> own_u) & 1u) != 0;
                read_group((uint32_t)(wid * GPW + owner + own_u * WAVE), own_dirty, wtop, fl);
                if (lane == owner) q_ub[own_u][tid] = 0;
            }
#ifndef DA_AB_NO_IDLE_REPAIR
            else {  // nothing of its own to read: one of the groups with a stale bound, if there are any
                unsigned int k = 0;
                if (lane == 0) k = atomicAdd(&q_stake, 1u);
                k = (unsigned int)__builtin_amdgcn_readfirstlane((int)k);
                if (k < q_sn) {
                    const uint32_t at = q_sl[k];
                    const int u = (int)(at / SEL2_THREADS), t = (int)(at % SEL2_THREADS);
                    read_group((uint32_t)((t / WAVE) * GPW + (t % WAVE) + u * WAVE), true, 0ull, fl);
                }
            }
#endif
        }
        __syncthreads();
        // ---- the groups that still reach the floor, from all waves into one work list, dealt out evenly: a wave that owns several of them
        // no longer reads them one after the other while the others idle (the launch lasts as long as its slowest wave)
        {
            const unsigned long long fl = __hip_atomic_load(&q_floor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int u = 0; u < GPL; ++u) {
                const unsigned long long v = q_ub[u][tid];
                if (v != 0 && v >= fl && !(rmask >> u & 1u)) q_work[atomicAdd(&q_wn, 1u)] = (uint32_t)(u * SEL2_THREADS + tid) | (((dmask >> u) & 1u) << 31);
            }
        }
        __syncthreads();
        {
            const unsigned int wn = q_wn;
#ifdef DA_PHASE_TIMERS
            if (tid == 0) {
                if (wn > 32) g->st_qdiag[5] += 1;
                if (wn > g->st_qdiag[6]) g->st_qdiag[6] = wn;
                if (pass == 0) g->st_qdiag[7] += 1;
                g->st_qdiag[8] += wn;
            }
#endif
            // the work list, then the stale groups round 0 left: those ride in the rounds the work list needs anyway (waves that would idle), beyond
            // that only the ones whose bound reaches the floor are read now
            const unsigned int sn = q_sn, taken = min(q_stake, sn), rounds = max((wn + NW - 1) / NW, min((unsigned int)DA_AB_REPAIR_ROUNDS, (sn - taken) / 48u));  // (a long backlog -- the first steps of a chain -- may get rounds of its own)
            for (unsigned int i = (unsigned int)wid; i < wn + (sn - taken); i += NW) {  // (wave-uniform)
                const bool listed = i < wn;
                const uint32_t info = listed ? q_work[i] : q_sl[taken + (i - wn)], at = info & 0x7FFFFFFFu;
                const int u = (int)(at / SEL2_THREADS), t = (int)(at % SEL2_THREADS);
                const unsigned long long bound = q_ub[u][t];
                const unsigned long long fl = __hip_atomic_load(&q_floor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (bound < fl && (listed || i / NW >= rounds)) continue;  // the floor has risen past it meanwhile / a stale group that can wait
                read_group((uint32_t)((t / WAVE) * GPW + (t % WAVE) + u * WAVE), !listed || (info >> 31) != 0, bound, fl);
#ifdef DA_PHASE_TIMERS
                if (pass) ++q_rounds;
#endif
            }
        }
        // the wave's best entry outside the excluded rows (highest rank, then highest tie word among its holders)
        const uint32_t wnr = wave_max_u32(nr);
        const unsigned long long wnt = wave_max_u64(nr == wnr ? nt : 0ull);
        // every wave fetches list references and records of ITS candidate's rows now (in flight across the reduction)
        const uint32_t cA = wnr ? (uint32_t)((wnt >> 7) & 0xFFFFFFu) : 0u, cB = wnr ? (uint32_t)(wnt >> 31) : 0u;
        const RowInfo cand_ra = load_row(c.rows, cA), cand_rb = load_row(c.rows, cB);
        const da_u2 cand_refA = rowoff[cA], cand_refB = rowoff[cB];
        if (lane == 0) {
            q_red_rank[wid] = wnr;
            q_red_tie[wid] = wnt;
        }
        __syncthreads();
        uint32_t best_rank;
        unsigned long long best_tie;
        {
            const uint32_t r = lane < NW ? q_red_rank[lane] : 0u;
            const unsigned long long t = lane < NW ? q_red_tie[lane] : 0ull;
            best_rank = wave_max_u32(r);
            best_tie = wave_max_u64(r == best_rank ? t : 0ull);
        }
        const bool mine = best_rank != 0 && wnr == best_rank && wnt == best_tie;  // exactly one wave holds the winner (tie words are unique)
        if (pass == 0) {
            Q_TIMER_MARK(2)
            // (the substitution block of this launch finds the same pick by itself -- table_argmax_block: the two blocks never wait for each other)
            if (best_rank == 0) {
                if (tid == 0) {
                    g->spec[par].word = 0;
                    g->l_n[par] = 0;
                }
                if (lane == 0 && rescans) atomicAdd(&g->st_rescans, (unsigned long long)rescans);
                return;  // the table holds nothing selectable: the chain ends
            }
            exA = (uint32_t)((best_tie >> 7) & 0xFFFFFFu);
            exB = (uint32_t)(best_tie >> 31);
            __threadfence();  // the bounds this pass tightened are re-read by the next one: stores complete, this CU's L1 dropped
            __syncthreads();
        } else {
            // R for the next step, and with it the entries of the pick's rows that reach it (all scans ended before the barrier above: q_Ln is final)
            const unsigned long long r0 = best_rank ? bound_word(best_rank, best_tie) : 0ull;
            const unsigned int met = q_Ln;
            if (best_rank != 0 && met != 0 && met <= (unsigned int)QL_CAP) {  // (block-uniform)
                if (tid < (int)met) {
                    const CandEntry e = q_L[tid];
                    if (bound_word((uint32_t)e.rank, e.tie) >= r0) {
                        const unsigned int at = atomicAdd(&q_Lout, 1u);
                        if (at < (unsigned int)TOUCH_CAP) {
                            ll_out[at].rank = e.rank;
                            ll_out[at].tie = e.tie;
                        }
                    }
                }
                __syncthreads();
            }
            const unsigned int kept = q_Lout;
            const bool usable = best_rank != 0 && met <= (unsigned int)QL_CAP && kept <= (unsigned int)TOUCH_CAP;
            if (best_rank == 0) {
                if (tid == 0) {
                    g->spec[par].word = 0;
                    g->l_n[par] = 0;
                }
            } else if (mine && lane == 0) {
                SpecPick sp;
                sp.word = usable ? r0 : 0ull;
                sp.tie = best_tie;
                sp.refA = cand_refA;
                sp.refB = cand_refB;
                sp.ra = cand_ra;
                sp.rb = cand_rb;
                g->spec[par] = sp;
                g->l_n[par] = usable ? kept : 0u;
            }
        }
    }
    Q_TIMER_MARK(3)
    if (lane == 0 && rescans) atomicAdd(&g->st_rescans, (unsigned long long)rescans);
    if (tid == 0) {
#ifdef DA_PHASE_TIMERS
        if (DA_TIMED_STEP(step)) {
            g->st_qphase[0] += (unsigned long long)(qp[1] - qp[0]);
            if (!fast) g->st_qphase[1] += (unsigned long long)(qp[2] - qp[1]);
            g->st_qphase[2] += (unsigned long long)(qp[3] - (fast ? qp[1] : qp[2]));
            g->st_qphase[3] += 1;
        }
#endif
    }
#ifdef DA_PHASE_TIMERS
    if (lane == 0) {  // per wave: re-reads that found the bound stale and below the floor / of clean groups for an excluded best entry; the longest wave's rounds
        atomicAdd(&g->st_qdiag[0], (unsigned long long)q_stale);
        atomicAdd(&g->st_qdiag[1], (unsigned long long)q_touch);
        atomicMax(&g->st_qdiag[2 + (step & 1)], (unsigned long long)q_rounds);  // (max over the waves of this step; folded by the next-but-one step below)
    }
    if (tid == 0) {
        g->st_qdiag[4] += g->st_qdiag[2 + ((step + 1) & 1)];
        g->st_qdiag[2 + ((step + 1) & 1)] = 0;
    }
#endif
}

// pick_body: the substitution block of a step (see above): substitution of the pick in the
// lists of rows A and B, exact recount of the six pairs among {A, B, new} (their counts go to sp_cnt: the blocks are written by
// k_iter_update), partner rows -- behind a pick that is either read from the descriptor (known one step ahead) or found by the block itself
// (table_argmax_block).  SHARDED (cmvm_shard.h): the chain holds a slice of the columns and a replica of the pair table -- the six count vectors are
// PARTIAL and go to the head of the exchange slab, and the block leaves one flag per row that shares a substituted column instead of a partner list.
// Returns 1 when the chain is (or just became) finished.
template <class Cell, bool SHARDED = false> __device__ __forceinline__ int pick_body(ChainDev *g, unsigned int *n_done, int step) {
    using O = CellOps<Cell>;
    using F = RowFmt<Cell>;
    using Entry = typename F::Entry;
    const int par = step & 1, iter = step;
    // ---- ONE scalar round trip: every descriptor field the block needs, the entry left by the previous step's search included
    int was_done = g->done, had_error = g->error, lcap = g->lcap, claim_words = g->claim_words;
    int n_rows0 = g->n_rows, rcap = g->rcap, cbw = g->cb_words, adder_size = g->adder_size, carry_size = g->carry_size;
    uint32_t offN = g->rl_used, rl_cap = g->rl_cap, n_live0 = g->n_live, live_peak0 = g->live_peak;
    const uint32_t *step_mant = g->step_mant;
    const float *step_tab = g->step_tab;
    int n_step_mant = g->n_step_mant;
    unsigned long long sp_word = g->spec[par ^ 1].word, sp_tie = g->spec[par ^ 1].tie;
    unsigned int cn_prev = g->c_n[par ^ 1];
    const DA_GLOBAL CandEntry *cl_prev = (const DA_GLOBAL CandEntry *)&g->c_list[par ^ 1][0];
    uint32_t sp_ax = g->spec[par ^ 1].refA.x, sp_ay = g->spec[par ^ 1].refA.y, sp_bx = g->spec[par ^ 1].refB.x, sp_by = g->spec[par ^ 1].refB.y;
    float sp_ra0 = g->spec[par ^ 1].ra.lo, sp_ra1 = g->spec[par ^ 1].ra.hi, sp_ra2 = g->spec[par ^ 1].ra.step, sp_ra3 = g->spec[par ^ 1].ra.lat;
    float sp_rb0 = g->spec[par ^ 1].rb.lo, sp_rb1 = g->spec[par ^ 1].rb.hi, sp_rb2 = g->spec[par ^ 1].rb.step, sp_rb3 = g->spec[par ^ 1].rb.lat;
    Ctx c = make_ctx_raw(g, 2 * iter);
    DA_GLOBAL int *collen = (DA_GLOBAL int *)g->collen;
    DA_GLOBAL da_u2 *rowoff = (DA_GLOBAL da_u2 *)g->rowoff;
    DA_GLOBAL Entry *rl = (DA_GLOBAL Entry *)g->rlist;
    DA_GLOBAL Cell *mA = (DA_GLOBAL Cell *)g->mA, *mB = (DA_GLOBAL Cell *)g->mB;
    DA_GLOBAL int *mcol = (DA_GLOBAL int *)g->mcol;
    DA_GLOBAL unsigned long long *collist = (DA_GLOBAL unsigned long long *)g->collist;
    DA_GLOBAL uint16_t *cmap = (DA_GLOBAL uint16_t *)g->cmap;
    DA_GLOBAL uint32_t *colbits = (DA_GLOBAL uint32_t *)g->colbits;
    DA_GLOBAL uint32_t *pl_ids = (DA_GLOBAL uint32_t *)g->pl_ids;
    DA_GLOBAL unsigned long long *plist = (DA_GLOBAL unsigned long long *)g->plist;
    DA_GLOBAL da_i4 *picks = (DA_GLOBAL da_i4 *)g->picks;
    DA_GLOBAL uint32_t *sp_cnt = (DA_GLOBAL uint32_t *)g->sp_cnt;
    DA_GLOBAL int32_t *cs_slab = (DA_GLOBAL int32_t *)g->cs_slab, *cs_flags = (DA_GLOBAL int32_t *)g->cs_flags;  // (column-sharded chains only)
    pin_sgpr(was_done, had_error, lcap, claim_words, n_rows0, rcap, cbw, adder_size, carry_size, offN, rl_cap, n_live0, live_peak0, step_mant, step_tab, n_step_mant);
    pin_sgpr(sp_word, sp_tie, cn_prev, cl_prev, sp_ax, sp_ay, sp_bx, sp_by, sp_ra0, sp_ra1, sp_ra2, sp_ra3, sp_rb0, sp_rb1, sp_rb2, sp_rb3);
    pin_sgpr(c.n_out, c.n_bits, c.K, c.Kpad, c.rows);
    pin_sgpr(collen, rowoff, rl, mA, mB, mcol, collist, cmap, colbits, pl_ids, plist, picks, sp_cnt);
    if constexpr (SHARDED) pin_sgpr(cs_slab, cs_flags);
    const int n_out = c.n_out, Kpad = c.Kpad, nb = c.n_bits;
    // column-sharded chain (cmvm_shard.h): a rank whose chain stops (finished, or an error) still hands out the flag buffer of the step -- zero
    // flags and the status trailer {1, capacity error?, other error?} -- written here, on the device: the host does not read the descriptor
    // back between the steps
    auto shard_stop = [&](int err) {
        if constexpr (SHARDED) {
            const int fw = (n_rows0 + 3) / 4;  // rows before this step
            for (int w = threadIdx.x; w < fw; w += SEL2_THREADS) cs_flags[w] = 0;
            if (threadIdx.x == 0) {
                const bool cap = err == E_ROW_CAPACITY || err == E_TABLE_CAPACITY || err == E_LIST_CAPACITY;
                cs_flags[fw] = 1;
                cs_flags[fw + 1] = cap ? 1 : 0;
                cs_flags[fw + 2] = err != E_OK && !cap ? 1 : 0;
            }
        }
    };
    if (was_done) {
        shard_stop(had_error);
        return 1;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // dynamic LDS carve: B's list | special-pair counters | per-matched-column scratch | column -> position in B
    Entry *s_bent = reinterpret_cast<Entry *>(smem);                                  // [n_out] entries of row B (updated in place)
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_bent + n_out);                  // [6][Kpad]
    int *s_len = reinterpret_cast<int *>(s_cnt + 6 * Kpad);                           // [n_out + 1] list lengths of the matched columns
    int *s_col = s_len + n_out + 1;                                                   // [n_out] matched columns
    int *s_bpos = s_col + n_out;                                                      // [n_out] 1 + position of a column in B's list, 0 = absent
    int *s_clen = s_bpos + n_out;                                                     // [n_out] list length of every column
    int *s_cm = s_clen + n_out;                                                       // [n_out] 1 + index among the matched columns, 0 = not matched
    uint32_t *s_or = reinterpret_cast<uint32_t *>(s_cm + n_out);                      // [claim_words] OR of the substituted columns' row bitmaps (if it fits)
    constexpr int NW = SEL2_THREADS / WAVE;
    __shared__ int s_np, s_part[NW];
    __shared__ unsigned int s_matches;
    __shared__ RowInfo s_new;
    __shared__ unsigned long long s_am_tie[NW];  // table_argmax_block (a step whose pick was not known ahead)
    __shared__ uint32_t s_am_rank[NW];
    constexpr int IDS_LDS = DA_IDS_LDS;
    __shared__ uint32_t s_ids[IDS_LDS];  // the first partner row ids of the step (the rest, if any, goes through pl_ids in HBM)
    __shared__ Log2Table s_log2;  // copy of c_log2 (the latency model's look-up then stays off the memory path)

    const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();

    SEL_TIMER_DECL
    SEL_TIMER_MARK(0)
    if (had_error != E_OK) {  // a capacity error poisons the chain: stop it (the host retries with a larger arena)
        if (tid == 0) {
            g->done = 1;
            g->n_partners = 0;
            atomicAdd(n_done, 1u);
        }
        shard_stop(had_error);
        return 1;
    }
    // ---------------- (1) the pick: the best of what the previous step left (its untouched entry against the entries its update listed), or --
    // when it left nothing usable -- awaited from the search block of this launch
    unsigned long long best_tie;
    bool from_spec;
    const bool fast = resolve_pick(cl_prev, sp_word, sp_tie, cn_prev, best_tie, from_spec);
    uint32_t A = (uint32_t)((best_tie >> 7) & 0xFFFFFFu), B = (uint32_t)(best_tie >> 31);
    da_u2 refA = da_u2{sp_ax, sp_ay}, refB = da_u2{sp_bx, sp_by};
    RowInfo ra = RowInfo{sp_ra0, sp_ra1, sp_ra2, sp_ra3}, rb = RowInfo{sp_rb0, sp_rb1, sp_rb2, sp_rb3};
    if (fast && !from_spec) {  // (block-uniform) a listed entry won: its rows' list references and records (one round trip, broadcast loads)
        refA = rowoff[A];
        refB = rowoff[B];
        ra = load_row(c.rows, A);
        rb = load_row(c.rows, B);
    }
    {
        // the list lengths of all columns (needed for the matched columns only, after the substitution) and the latency model's table
        const int clen0 = collen[tid < n_out ? tid : 0];
        const bool want_log2 = (adder_size >= 0 || carry_size >= 0) && tid < (int)(sizeof(Log2Table) / 4);
        const uint32_t l2w = want_log2 ? reinterpret_cast<const uint32_t *>(&c_log2)[tid] : 0u;
        if (tid == 0) {
            s_np = 0;
            s_matches = 0;
            g->c_n[par] = 0;  // this step's update appends to c_list[par] (last read by the selection of step - 1)
        }
        if (tid < n_out) {
            s_clen[tid] = clen0;
            s_cm[tid] = 0;
            s_bpos[tid] = 0;
        }
        if (want_log2) reinterpret_cast<uint32_t *>(&s_log2)[tid] = l2w;
        for (int w = tid; w < claim_words; w += SEL2_THREADS) s_or[w] = 0;
        for (int j = tid + SEL2_THREADS; j < n_out; j += SEL2_THREADS) {
            s_clen[j] = collen[j];
            s_cm[j] = 0;
            s_bpos[j] = 0;
        }
    }
    SEL_TIMER_MARK(1)
    unsigned long long pk_word = 1;
    if (!fast) {
        // (block-uniform, rare: the first step of a chain, an overflowed list) no pick known ahead: the arg-max of the table by this block itself, a
        // plain pass over the rank array -- nothing is written, no bound is read, and nothing is awaited from the search block of this launch, which
        // finds the same entry through the bounds for its own purpose (HIP promises nothing about the order in which workgroups are dispatched)
        uint32_t r0;
        table_argmax_block(g, r0, best_tie, s_am_rank, s_am_tie);
        pk_word = r0 ? bound_word(r0, best_tie) : 0ull;
        A = (uint32_t)((best_tie >> 7) & 0xFFFFFFu);
        B = (uint32_t)(best_tie >> 31);
        if (pk_word) {
            refA = rowoff[A];
            refB = rowoff[B];
            ra = load_row(c.rows, A);
            rb = load_row(c.rows, B);
        }
    }
    __syncthreads();
    SEL_TIMER_MARK(2)
    const uint32_t Nw = (uint32_t)n_rows0;
    if (pk_word == 0 || (int)Nw >= rcap) {
        if (tid == 0) {
            if (pk_word != 0) g->error = E_ROW_CAPACITY;
            g->done = 1;
            g->n_partners = 0;
            atomicAdd(n_done, 1u);
        }
        shard_stop(pk_word != 0 ? E_ROW_CAPACITY : E_OK);
        return 1;
    }
    const int idx = (int)(best_tie & 0x7F);
    int shift, sub;
    key_decode(idx, nb, shift, sub);
    const bool same = A == B;

    // ---------------- (2) new row record + substitution (as select_body)
    const int lenA = (int)refA.y, lenB = (int)refB.y;
    if (offN + (uint32_t)lenA > rl_cap) {  // the new row has at most lenA entries (cannot happen with the exact bound; guarded)
        if (tid == 0) {
            g->error = E_LIST_CAPACITY;
            g->done = 1;
            g->n_partners = 0;
            atomicAdd(n_done, 1u);
        }
        shard_stop(E_LIST_CAPACITY);
        return 1;
    }
    DA_GLOBAL Entry *rlA = rl + refA.x, *rlB = rl + refB.x, *rlN = rl + offN;
    Entry eA0 = rl[tid < lenA ? refA.x + (uint32_t)tid : 0u], eB0 = rl[tid < lenB ? refB.x + (uint32_t)tid : 0u];
    load_fence();
    RowInfo rn = RowInfo{0.0f, 0.0f, 0.0f, 0.0f};
    int derr = 0;
    if (tid == 0) {  // arithmetic only, while the entries are in flight; the global stores follow once they have been consumed
        qint_add_pair(ra, rb, shift, sub, rn.lo, rn.hi, rn.step);
        float dlat = adder_dlat(ra, rb, shift, sub, adder_size, carry_size, s_log2, StepLog2{n_step_mant, step_mant, step_tab}, derr);
        rn.lat = (ra.lat < rb.lat ? rb.lat : ra.lat) + dlat;
        s_new = rn;
    }
    for (int k = tid; k < 6 * Kpad; k += SEL2_THREADS) s_cnt[k] = 0;
    if (tid >= lenA) eA0 = F::none();
    if (same || tid >= lenB) eB0 = F::none();
    pin_vgpr(eA0, eB0);  // both consumed (waited for) here, in front of thread 0's stores below
    if (!same) {  // B's list into LDS, addressable by column (s_bpos was zeroed in the prologue, a barrier ago)
        if (tid < lenB) {
            s_bent[tid] = eB0;
            s_bpos[F::col(eB0)] = tid + 1;
        }
        for (int t = tid + SEL2_THREADS; t < lenB; t += SEL2_THREADS) {
            const Entry e = rlB[t];
            s_bent[t] = e;
            s_bpos[F::col(e)] = t + 1;
        }
    }
    if (tid == 0) {  // (a wait for a loaded value also waits for every store issued before it: these come after the last one)
        if (derr) g->error = E_FLOAT_DOMAIN;
        store_row((DA_GLOBAL RowInfo *)c.rows, Nw, rn);
        picks[iter] = da_i4{(int)A, (int)B, sub, shift};
        if (n_live0 > live_peak0) g->live_peak = n_live0;
    }
    __syncthreads();
    SEL_TIMER_MARK(3)
    uint32_t *cAA = s_cnt, *cAB = s_cnt + Kpad, *cBB = s_cnt + 2 * Kpad, *cAN = s_cnt + 3 * Kpad, *cBN = s_cnt + 4 * Kpad,
             *cNN = s_cnt + 5 * Kpad;
    unsigned int my_matches = 0;
    int m = 0;  // matched columns so far (block-uniform)
    // pass 1: one thread per entry of A (ascending columns); the matched columns are compacted in column order, which
    // is the order of the new row's list
    for (int t0 = 0; t0 < lenA; t0 += SEL2_THREADS) {
        const int t = t0 + tid;
        Cell a = 0, b = 0, ma = 0, mb = 0, na = 0, nbv = 0;
        uint32_t colA = 0;
        int pos = 0;
        if (t < lenA) {
            const Entry e = t0 == 0 ? eA0 : rlA[t];
            colA = F::col(e);
            a = F::cell(e);
            if (same)
                b = a;
            else {
                pos = s_bpos[colA];
                b = pos ? F::cell(s_bent[pos - 1]) : (Cell)0;
            }
            if (a && b) substitute_column<Cell>(a, b, same, shift, sub, ma, mb);
            na = same ? (Cell)(a & ~ma & ~mb) : (Cell)(a & ~ma);
            nbv = b & ~mb;
        }
        const bool hit = ma != 0;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_part[wid] = __popcll(bal);
        __syncthreads();
        int wbase = 0, chunk = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int v = s_part[w];
            wbase += w < wid ? v : 0;
            chunk += v;
        }
        if (hit) {
            const int at = m + wbase + __popcll(bal & ((1ull << lane) - 1));
            rlA[t] = F::pack(colA, na);
            if (!same) s_bent[pos - 1] = F::pack(colA, nbv);
            rlN[at] = F::pack(colA, ma);
            mcol[at] = (int)colA;
            mA[at] = ma;
            mB[at] = mb;
            s_cm[colA] = at + 1;
            {  // row bitmaps of this column: the new row enters, a row whose cell just lost its last digit leaves
                DA_GLOBAL uint32_t *cb = colbits + (size_t)colA * cbw;
                atomicOr(gen(&cb[Nw >> 5]), 1u << (Nw & 31));
                if (na == 0) atomicAnd(gen(&cb[A >> 5]), ~(1u << (A & 31)));
                if (!same && nbv == 0) atomicAnd(gen(&cb[B >> 5]), ~(1u << (B & 31)));
            }
            s_len[at] = s_clen[colA];  // the pre-append length: the new row itself is not a partner
            s_col[at] = (int)colA;
            my_matches += popc32(O::plus(ma) | O::minus(ma));
        }
        // ---------------- (3) exact recount of the pairs among {A, B, new} (their old blocks are replaced); the
        // self pairs of B are counted in pass 2 (B may have columns A does not have)
        if (na) for_pairs_self<Cell>(na, nb, [&](int k) { atomicAdd(&cAA[k], 1u); });
        if (!same) {
            if (na && nbv) for_pairs_cross<Cell>(na, nbv, nb, [&](int k) { atomicAdd(&cAB[k], 1u); });
            if (nbv && ma) for_pairs_cross<Cell>(nbv, ma, nb, [&](int k) { atomicAdd(&cBN[k], 1u); });
        }
        if (na && ma) for_pairs_cross<Cell>(na, ma, nb, [&](int k) { atomicAdd(&cAN[k], 1u); });
        if (ma) for_pairs_self<Cell>(ma, nb, [&](int k) { atomicAdd(&cNN[k], 1u); });
        m += chunk;
        __syncthreads();  // s_part is reused by the next chunk; s_bent updates visible to pass 2
    }
    if (my_matches) atomicAdd(&s_matches, my_matches);  // LDS; added to the chain's statistics by thread 0 at the end
    // pass 2: B's list back to memory, self pairs of what is left of B
    if (!same)
        for (int t = tid; t < lenB; t += SEL2_THREADS) {
            const Entry e = s_bent[t];
            rlB[t] = e;
            const Cell nbv = F::cell(e);
            if (nbv) for_pairs_self<Cell>(nbv, nb, [&](int k) { atomicAdd(&cBB[k], 1u); });
        }
    // the map column -> matched index for the update blocks (one coalesced copy; pass 1 has completed: its last barrier)
    for (int j = tid; j < n_out; j += SEL2_THREADS) cmap[j] = (uint16_t)s_cm[j];
    // the new row joins the lists of its columns
    {
        const unsigned long long refN = ref_pack(Nw, (uint32_t)m, offN);
        for (int k = tid; k < m; k += SEL2_THREADS) {
            const int j = s_col[k], len = s_len[k];
            if (len < lcap) {
                collist[(size_t)j * lcap + len] = refN;
                collen[j] = len + 1;
            } else
                g->error = E_LIST_CAPACITY;
        }
    }
    __syncthreads();
    SEL_TIMER_MARK(4)
    SEL_TIMER_MARK(5)
    // the exact counts of the six pairs among {A, B, new}: to the special-pair block of k_iter_update (all counters are final: the barrier above)
    if constexpr (SHARDED) {  // PARTIAL counts (this rank's columns): the head of the exchange slab, [6][K]; pairs with B do not exist when the pick is a row with itself
        for (int k = tid; k < 6 * c.K; k += SEL2_THREADS) {
            const int sp = k / c.K;
            cs_slab[k] = (same && (sp == 1 || sp == 2 || sp == 4)) ? 0 : (int32_t)s_cnt[sp * Kpad + (k - sp * c.K)];
        }
    } else
        for (int k = tid; k < 6 * Kpad; k += SEL2_THREADS) sp_cnt[k] = s_cnt[k];
    // ---------------- (4) partner rows -- the rows that have digits in a substituted column -- into the partner list:
    // OR of those columns' row bitmaps (A, B and the new row masked out: their bits are being changed by this very kernel).  Word w
    // of the OR covers rows 32 w ..; the set bits are counted (DPP prefix sum), one LDS atomic per wave reserves the places, the
    // row ids stay in LDS; then, one thread per partner, the list reference is attached (a parallel gather from rowoff).
    {
        const DA_GLOBAL uint32_t *cb = colbits;
        const int nwords = (int)((Nw + 31) >> 5), chunks = (nwords + WAVE - 1) / WAVE;
        DA_GLOBAL uint32_t *ids = pl_ids;
        // While the chain is young the bitmaps are a few 64-word chunks long and many columns are substituted (tens): one wave per chunk
        // would OR them one load after the other while the others idle (20 k of the 34 k cycles of an early step, measured).  The waves of
        // a chunk share its columns then and combine their parts in LDS.
        const bool split = chunks * 4 <= NW && m >= 8 && nwords <= claim_words;  // (at least four waves per chunk and two columns each: else the barrier costs more than it saves)
        if (split) {
            const int c = wid % chunks, part = wid / chunks, ways = (NW - c + chunks - 1) / chunks;  // waves c, c + chunks, ... serve chunk c
            const int w = c * WAVE + lane;
            if (w < nwords) {
                uint32_t bits = 0;
                for (int k = part; k < m; k += ways) bits |= cb[(size_t)s_col[k] * cbw + w];
                if (bits) atomicOr(&s_or[w], bits);
            }
            __syncthreads();
        }
        for (int wb = wid * WAVE; wb < nwords; wb += SEL2_THREADS) {  // wave-uniform trip count
            const int w = wb + lane;
            uint32_t bits = 0;
            if (w < nwords) {
                if (split)
                    bits = s_or[w];
                else
                    for (int k = 0; k < m; ++k) bits |= cb[(size_t)s_col[k] * cbw + w];
                if ((int)(A >> 5) == w) bits &= ~(1u << (A & 31));
                if ((int)(B >> 5) == w) bits &= ~(1u << (B & 31));
                if ((int)(Nw >> 5) == w) bits &= ~(1u << (Nw & 31));
            }
            if constexpr (SHARDED) {  // one 8-bit flag field per row (four rows per word: summed over the ranks without carry) instead of a partner list
                const int fw = (int)((Nw + 3) / 4);
                if (w < nwords) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const uint32_t b = (bits >> (4 * q)) & 0xFu;
                        if (8 * w + q < fw) cs_flags[8 * w + q] = (int32_t)((b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21));
                    }
                }
                continue;
            }
            const int cnt = popc32(bits), inc = (int)wave_scan_add_u32((uint32_t)cnt);  // inclusive prefix inside the wave (DPP)
            const int wave_total = __builtin_amdgcn_readlane(inc, WAVE - 1);
            if (wave_total == 0) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_np, wave_total);
            int at = __builtin_amdgcn_readfirstlane(base) + inc - cnt;
            while (bits) {
                const uint32_t id = (uint32_t)(w << 5) + (uint32_t)ctz32(bits);
                if (at < IDS_LDS)
                    s_ids[at] = id;
                else
                    ids[at] = id;
                ++at;
                bits &= bits - 1;
            }
        }
    }
    SEL_TIMER_MARK(6)
    __syncthreads();
    if constexpr (SHARDED) {
        if (tid < SHARD_TRAILER) cs_flags[(int)((Nw + 3) / 4) + tid] = 0;  // status trailer of a rank that took the step (cmvm_shard.h)
    }
    {  // partner ids -> partner list entries (row id, list length, list offset)
        const int np = s_np;
        const DA_GLOBAL uint32_t *ids = pl_ids;
        for (int t = tid; t < np; t += 2 * SEL2_THREADS) {  // two partners per thread and pass, their look-ups in flight together
            const int t2 = t + SEL2_THREADS;
            const bool has2 = t2 < np;
            const uint32_t r1 = t < IDS_LDS ? s_ids[t] : ids[t];
            const uint32_t r2 = !has2 ? r1 : t2 < IDS_LDS ? s_ids[t2] : ids[t2];
            const da_u2 ro1 = rowoff[r1], ro2 = rowoff[r2];
            load_fence();
            plist[t] = ref_pack(r1, ro1.y, ro1.x);
            if (has2) plist[t2] = ref_pack(r2, ro2.y, ro2.x);
        }
    }
    SEL_TIMER_MARK(7)
    if (tid == 0) {
        SEL_TIMER_FLUSH
        rowoff[Nw] = da_u2{offN, (uint32_t)m};
        g->rl_used = offN + (uint32_t)m;
        g->m = m;
        g->n_partners = s_np;
        // statistics: atomics without a return value (a plain += is load -> add -> store: one more round trip at the very end)
        atomicAdd(&g->st_matches, (unsigned long long)s_matches);
        atomicAdd(&g->st_partners, (unsigned long long)s_np);
        atomicAdd(&g->st_cells, (unsigned long long)s_np * (unsigned)m);
        if (fast) atomicAdd(&g->st_fast, 1ull);
        {
            // algorithmic bytes of this step's substitution block: the list lengths of all columns, the pick (64 B),
            // both row lists read and written back, the row bitmaps of the m substituted columns, per partner its id + list reference +
            // partner-list entry, the hand-off (matched columns, consumed digits, new row's list, column map, column-list and bitmap
            // updates), the six special-pair count vectors.  (The search block's bytes: st_rescans, and 17 B per group, priced by the host.)
            const unsigned long long eb = sizeof(Entry), cb = sizeof(Cell);
            const unsigned long long nwords = (Nw + 31) >> 5;
            atomicAdd(&g->st_sel_bytes, 4ull * (unsigned)n_out + 64ull + 2ull * eb * (unsigned)(lenA + (same ? 0 : lenB)) + 4ull * (unsigned)m * nwords + 20ull * (unsigned)s_np +
                                            (unsigned)m * (4ull + 2ull * cb + eb + 8ull + 12ull) + 2ull * (unsigned)n_out + 6ull * 4ull * (unsigned)Kpad);
        }
        g->A = A;
        g->B = B;
        g->Nw = Nw;
        g->pk_shift = shift;
        g->pk_sub = sub;
        g->n_rows = (int)Nw + 1;
        g->iter = iter + 1;
    }
    return 0;
}
// grid (chains padded to a multiple of 8, 2): the chain index is the fast grid dimension, so that the two blocks of a chain and its
// k_iter_update blocks run on the same XCD (update_body); y = 0 the search block, y = 1 the substitution block.  `step` = the
// number of the lockstep iteration = the `iter` of every chain of the launch that has not finished (kernel argument: the search
// block must not read a field the substitution block writes during the launch).
#ifndef DA_SEL2_WAVES
#define DA_SEL2_WAVES 4  // wavefronts per SIMD the register budget of k_iter_select2 is capped for (measured 4 .. 8: the spills of 5 and more cost more than the smaller footprint gains)
#endif
template <class Cell, bool SHARDED = false> __global__ void __launch_bounds__(SEL2_THREADS) __attribute__((amdgpu_waves_per_eu(DA_SEL2_WAVES, DA_SEL2_WAVES))) k_iter_select2(ChainDev *chains, int n_chains, unsigned int *n_done, int step) {
    if ((int)blockIdx.x >= n_chains) return;
#ifdef DA_STEP_CLOCKS
    {
        ChainDev *g = &chains[blockIdx.x];
        if (threadIdx.x == 0 && !g->done) {
            CLK_MARK(g, step & 1, 0, true);
            if (blockIdx.y == 1 && step > 0) {  // the step before this one is complete: its four marks into the sums
                unsigned long long *k = g->clk[(step - 1) & 1];
                if (k[0] && k[1] && k[2] && k[3] && CLK_TIMED_STEP(step - 1)) {
                    g->st_qdiag[0] += k[1] - ~k[0];
                    g->st_qdiag[1] += ~k[2] - k[1];
                    g->st_qdiag[4] += k[3] - ~k[2];
                    g->st_qdiag[5] += CLK_NOW() - k[3];
                    g->st_qdiag[7] += 1;
                }
                k[0] = k[1] = k[2] = k[3] = 0;
            }
        }
    }
#endif
    if (blockIdx.y == 0)
        search_body<Cell>(&chains[blockIdx.x], step);
    else
        (void)pick_body<Cell, SHARDED>(&chains[blockIdx.x], n_done, step);
#ifdef DA_STEP_CLOCKS
    __syncthreads();
    if (threadIdx.x == 0) CLK_MARK(&chains[blockIdx.x], step & 1, 1, false);
#endif
}

#ifndef DA_UPD_OCC
#define DA_UPD_OCC 5  // blocks of 256 threads per CU the register budget is capped for (88 VGPRs, no spills)
#endif
#if DA_UPD_OCC >= 8
#define DA_UPD_SGPRS 80  // 800 SGPRs per SIMD: more than 80 per wave would cap the residency below 8 waves per SIMD
#elif DA_UPD_OCC == 7
#define DA_UPD_SGPRS 96
#else
#define DA_UPD_SGPRS 102
#endif
constexpr int SLOT_NONE = -1, SLOT_SLOW = -2;  // no such block / to be searched beyond its first bucket

// ------------------------------------------------------------------------------------------------ k_iter_update (quad)
// The greedy loop is bound by VALU issue, not by memory: one wavefront per partner row executed ~260 vector and ~185
// scalar instructions per partner with a handful of lanes doing useful work (PMC: profiles/r02_*).  Here a partner row is
// handled by a 16-LANE GROUP -- one DPP row -- and a wavefront works on FOUR partners at once with one instruction
// stream: a key bucket is 16 slots = one load per lane, a typical row list has <= 16 entries, the K <= 32 counts of a
// block are 16 words = one word (two counts) per lane, and the reductions stay inside the DPP row.  Everything that was
// wave-uniform per partner (slots, keys, hashes) is now a per-lane value, identical across the lanes of a group.
// Rare cases (key not in its first bucket, block creation) fall back to the wave-wide table functions, one group at a time.
#ifndef DA_UPD_QG
#define DA_UPD_QG 16  // lanes per partner row (16: four partners per wavefront; 8: eight -- every wave instruction then serves twice as many partners, but a row list
                      // takes twice the chunks and a bucket two slots per lane: measured (MI355X, round 6, C3 batch / one chain) 32.5 / 24.2 us per step against 30.6 / 20.3)
#endif
constexpr int QG = DA_UPD_QG, QN = WAVE / QG;  // lanes per partner, partners per wavefront
constexpr int QG_LOG2 = QG == 16 ? 4 : 3, SPL = (int)BUCKET / QG, HL = QG / 2;  // key slots of a bucket per lane; lanes per block in the re-evaluation
constexpr uint32_t QMASK = (1u << QG) - 1u, HMASK = (1u << HL) - 1u;
static_assert(QG == 16 || QG == 8, "a partner row is handled by one DPP row or half a row");

__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long v) {  // max over each 16-lane row, valid in lane 15 of the row
    v = dpp_max_u64<DPP_ROW_SHR1, 0xF>(v);
    v = dpp_max_u64<DPP_ROW_SHR2, 0xF>(v);
    v = dpp_max_u64<DPP_ROW_SHR4, 0xF>(v);
    v = dpp_max_u64<DPP_ROW_SHR8, 0xF>(v);
    return v;
}

constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141;  // quad_perm:[1,0,3,2], quad_perm:[2,3,0,1], lane i <-> 7 - i of its eight
template <int LANES> __device__ __forceinline__ unsigned long long part_row_max_u64(unsigned long long v) {  // max over each group of EIGHT / FOUR lanes, in all of them (butterfly steps)
    v = dpp_max_u64<DPP_QUAD_XOR1, 0xF>(v);
    v = dpp_max_u64<DPP_QUAD_XOR2, 0xF>(v);
    if constexpr (LANES == 8) v = dpp_max_u64<DPP_ROW_HALF_MIRROR, 0xF>(v);
    return v;
}

// ---- one update step of a chain, as the wave-uniform values its workers need (scalar registers)
template <class Cell> struct UpdStep {
    Ctx c;
    int done, n_partners, m, n_in;
    uint32_t A, B, Nw;
    int shift, sub;  // of the pick's key
    const DA_GLOBAL int *mcol;
    const DA_GLOBAL Cell *mA, *mB;
    const DA_GLOBAL uint16_t *cmap;
    const DA_GLOBAL typename RowFmt<Cell>::Entry *rl;
    const DA_GLOBAL unsigned long long *plist;
};
// ONE round trip: every descriptor field an update worker needs, pinned before the first branch (pin_sgpr)
template <class Cell> __device__ __forceinline__ UpdStep<Cell> load_upd_step(ChainDev *gq) {
    using Entry = typename RowFmt<Cell>::Entry;
    UpdStep<Cell> u;
    int iter = gq->iter;
    u.done = gq->done, u.n_partners = gq->n_partners, u.m = gq->m, u.n_in = gq->n_in;
    u.A = gq->A, u.B = gq->B, u.Nw = gq->Nw;
    u.shift = gq->pk_shift, u.sub = gq->pk_sub;
    u.c = make_ctx_raw(gq, 2 * iter - 1);
    // the step being applied is iter - 1 (the selection has counted it): its search left the best entry the step does not touch in
    // spec[(iter - 1) & 1]; the best entry of every block this launch writes that reaches it goes to c_list[(iter - 1) & 1] (fold_entry)
    u.c.rword = gq->spec[(iter - 1) & 1].word;
    u.c.cn = &gq->c_n[(iter - 1) & 1];
    u.c.cl = &gq->c_list[(iter - 1) & 1][0];
    u.mcol = (const DA_GLOBAL int *)gq->mcol;
    u.mA = (const DA_GLOBAL Cell *)gq->mA, u.mB = (const DA_GLOBAL Cell *)gq->mB;
    u.cmap = (const DA_GLOBAL uint16_t *)gq->cmap;
    u.rl = (const DA_GLOBAL Entry *)gq->rlist;
    u.plist = (const DA_GLOBAL unsigned long long *)gq->plist;
    pin_sgpr(u.done, u.n_partners, iter, u.m, u.n_in, u.A, u.B, u.Nw, u.shift, u.sub, u.mcol, u.mA, u.cmap, u.rl, u.plist);
    pin_sgpr(u.c.n_out, u.c.n_bits, u.c.K, u.c.Kpad, u.c.method, u.c.gs_log2, u.c.pb_log2, u.c.cmask, u.c.windows, u.c.hkey, u.c.hrank, u.c.hblk, u.c.grec, u.c.rows);
    pin_sgpr(u.c.rword, u.c.cn, u.c.cl);
    u.c.tomb = KEY_TOMB - (unsigned long long)((2 * iter - 1) & 3);
    ctx_finish(u.c);
    return u;
}
// LDS of an update worker: the hand-off of the selection (consumed digits of A and B [n_out each], the substituted columns
// [n_out], column -> 1 + index of the substituted column [n_out]) -- shared by the waves that copied it together -- and ONE
// wave's per-partner counters [QN][3][Kpad] + a spare vector [Kpad] (counts of blocks that do not exist)
template <class Cell> struct UpdLds {
    Cell *mA, *mB;
    int *col;
    uint16_t *cmap;
    uint32_t *cnt;
    static __device__ __forceinline__ size_t table_bytes(int n_out) { return (size_t)n_out * (2 * sizeof(Cell) + 4 + 2); }
    static __device__ __forceinline__ size_t wave_bytes(int Kpad) { return (size_t)(QN * 3 + 1) * Kpad * 4; }  // (+ one spare vector)
    __device__ __forceinline__ void carve(unsigned char *tables, unsigned char *counters, int n_out) {
        mA = reinterpret_cast<Cell *>(tables);
        mB = mA + n_out;
        col = reinterpret_cast<int *>(mB + n_out);
        cmap = reinterpret_cast<uint16_t *>(col + n_out);
        cnt = reinterpret_cast<uint32_t *>(counters);
    }
};
// the hand-off of the selection into LDS by `nthr` threads (t = this thread's index among them): one pass; the caller makes
// it visible (block barrier, or lds_fence() when one wave copies for itself)
template <class Cell> __device__ __forceinline__ void copy_handoff(const UpdStep<Cell> &u, const UpdLds<Cell> &s, int t, int nthr) {
    for (int j = t; j < u.c.n_out; j += nthr) s.cmap[j] = u.cmap[j];
    for (int j = t; j < u.m; j += nthr) {
        s.col[j] = u.mcol[j];
        s.mA[j] = u.mA[j];
    }
}

// update_partners: ONE WAVEFRONT works through the partner rows first + q, first + stride + q, ... < limit of the step (q = its
// four 16-lane groups); no block-level synchronisation inside.  `ref_next` = the (pre-fetched) reference of this group's first
// partner, `rnew` the record of the new row.
// STATS: tally the blocks found / created / deleted (benchmark instrumentation and the "peak pair blocks" statistic, DA4ML_HIP_STATS=1); the ballots, population
// counts and the scalar registers of the three counters are 1.8 % of the batch's step (measured: 27.9 -> 27.5 us, one chain 19.5 -> 19.0), so the product runs without them
template <class Cell, bool STATS>
__device__ __forceinline__ void update_partners(ChainDev *g, const UpdStep<Cell> &u, const UpdLds<Cell> &s, int first, int stride, int limit, unsigned long long ref_next,
                                                const RowInfo &rnew, unsigned int &found, unsigned int &inserts, unsigned int &deletes) {
    using F = RowFmt<Cell>;
    using Entry = typename F::Entry;
    constexpr int HW = ((sizeof(Cell) == 4 ? 24 : 60) + HL - 1) / HL;  // count words per lane (a block's words over its HL lanes): narrow layout Kpad / 2 <= 24 words, wide <= 60
    const Ctx &c = u.c;
    const uint32_t A = u.A, B = u.B, Nw = u.Nw;
    const int m = u.m, n_in = u.n_in, pk_shift = u.shift, pk_sub = u.sub;
    const DA_GLOBAL Entry *rl = u.rl;
    const DA_GLOBAL unsigned long long *plist = u.plist;
    Cell *s_mA = s.mA;
    int *s_col = s.col;
    uint16_t *s_cmap = s.cmap;
    uint32_t *s_cnt = s.cnt;
    const int nb = c.n_bits, Kpad = c.Kpad, K = c.K;
    const int lane = lane_id();
    const int q = lane >> QG_LOG2, l = lane & (QG - 1), qsh = q * QG;
    const bool same = A == B;
    uint32_t *dA = s_cnt + (size_t)q * 3 * Kpad, *dB = dA + Kpad, *cN = dB + Kpad;  // this group's counters
    const int KW = Kpad / 2;  // 32-bit words of counts per block (two u16 counts each)
    UPD_TIMER_DECL
    // partner p of the list: pass p / (QN total_waves), wave (p / QN) mod total_waves, group p mod QN -- all groups of all
    // waves are busy except in the last pass
#ifndef DA_UPD_CH
#define DA_UPD_CH 4  // measured on MI355X (C3 batch 64): 1 -> 52.7, 2 -> 53.8, 4 -> 54.5 solves/s
#endif
    constexpr int CH = DA_UPD_CH;  // list chunks (16 entries each) fetched together; longer lists continue in the loop below
    for (int base = first; base < limit; base += stride) {
        const int idx = base + q;
        const bool valid = idx < limit;
        // ---- round trip 1: the group's partner reference (the next pass's one is fetched now: off the critical path there)
        const unsigned long long ref = ref_next;
        ref_next = idx + stride < limit ? plist[idx + stride] : 0ull;
        const uint32_t pr = ref_row(ref), off = ref_off(ref);
        const bool dense = valid && (int)pr < n_in;  // dense input row: entry j is column j -- fetch the substituted columns only
        const int cnt = !valid ? 0 : dense ? m : (int)ref_len(ref);
        // ---- round trip 2: the two key buckets (16 slots = one per lane) and the head of the row list
        const uint32_t lA = min(A, pr), hA = max(A, pr), lB = min(B, pr), hB = max(B, pr);
        const unsigned long long keyA = pack_pair(lA, hA), keyB = pack_pair(lB, hB);
        const uint32_t baseA = (hash_pair(lA, hA) & ~(BUCKET - 1)) & c.cmask, baseB = (hash_pair(lB, hB) & ~(BUCKET - 1)) & c.cmask;
        unsigned long long kA[SPL], kB[SPL];  // (16 slots of a bucket over the group's lanes: one or two -- adjacent -- per lane)
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
            kA[t] = valid ? c.hkey[baseA + l * SPL + t] : KEY_TOMB;
            kB[t] = (valid && !same) ? c.hkey[baseB + l * SPL + t] : KEY_TOMB;
        }
        Entry e[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = l + u * QG;
            e[u] = j < cnt ? rl[(size_t)off + (dense ? s_col[j] : j)] : F::none();
        }
        for (int k = l; k < 3 * Kpad; k += QG) dA[k] = 0;  // while the loads are in flight
        // ---- resolve the probes (per group: 16 bits of the wave ballots)
        int sA = SLOT_NONE, sB = SLOT_NONE;
        {
            // which of a lane's slots holds the key (a key sits in one slot at most), and whether the bucket has an EMPTY slot
            bool emp_a = false, emp_b = false;
            int at_a = -1, at_b = -1;
#pragma unroll
            for (int t = 0; t < SPL; ++t) {
                at_a = kA[t] == keyA ? t : at_a;
                at_b = kB[t] == keyB ? t : at_b;
                emp_a |= kA[t] == KEY_EMPTY;
                emp_b |= kB[t] == KEY_EMPTY;
            }
            const uint32_t hitA = (uint32_t)(__ballot(at_a >= 0) >> qsh) & QMASK, empA = (uint32_t)(__ballot(emp_a) >> qsh) & QMASK;
            const uint32_t hitB = (uint32_t)(__ballot(at_b >= 0) >> qsh) & QMASK, empB = (uint32_t)(__ballot(emp_b) >> qsh) & QMASK;
            // (SPL == 2: the slot inside the hit lane's pair comes from that lane; all lanes take part in the exchange)
            int ta = 0, tb = 0;
            if constexpr (SPL > 1) {
                ta = __shfl(at_a, hitA ? qsh + ctz32(hitA) : lane);
                tb = __shfl(at_b, hitB ? qsh + ctz32(hitB) : lane);
            }
            if (valid) {
                sA = hitA ? (int)(baseA + (uint32_t)(ctz32(hitA) * SPL + ta)) : (empA ? SLOT_NONE : SLOT_SLOW);
                if (!same) sB = hitB ? (int)(baseB + (uint32_t)(ctz32(hitB) * SPL + tb)) : (empB ? SLOT_NONE : SLOT_SLOW);
            }
        }
        UPD_TIMER_MARK(1)  // reference + list + table probes
        // ---- round trip 3: the payload lines of the blocks that exist.  The two blocks of a partner -- (A, r) and (B, r) -- are handled SIDE BY SIDE:
        // the lower half of the group's lanes takes A's block, the upper half B's, so that one pass of the instruction stream below re-evaluates both
        // (round 6: one pass per block was twice the instructions; batch 31.1 -> 30.7 us per step); a lane holds the header of its block (16 bytes)
        // and every HL-th count word
        const int half = l / HL, hl = l % HL;
        const int sX = half ? sB : sA;
        const unsigned long long keyX = half ? keyB : keyA;
        const da_i4 z4 = da_i4{0, 0, 0, 0};
        const da_i4 hdX = sX >= 0 ? *reinterpret_cast<const DA_GLOBAL da_i4 *>(blk_ptr(c, sX)) : z4;
        uint32_t wX[HW];
#pragma unroll
        for (int u = 0; u < HW; ++u) {
            const int j = hl + u * HL;
            wX[u] = (sX >= 0 && j < KW) ? reinterpret_cast<const DA_GLOBAL uint32_t *>(blk_ptr(c, sX) + 16)[j] : 0u;
        }
        lds_fence();  // counters are zero
        // ---- digit pairs lost with A's / B's consumed digits and gained with the new row: a lane per list entry.
        // The digit pairs (partner digit, consumed A-role digit) are walked ONCE for all the blocks they count in (round 6; three walks,
        // one per block, were most of this phase's instructions):
        //   * gained with the new row, whose cell in this column IS ma (the partner is the row with the smaller id): key (q - p, signs);
        //   * lost with A: the same key when the partner is also below A, the mirrored one (shift negated) otherwise;
        //   * lost with the B-role digit that went with the A-role one: it sits `shift` positions further, its sign flipped when the pick
        //     subtracts (substitute_column, cmvm_core.h: mb = ma moved by shift, also when A and B are one row) -- in B's block, or in A's
        //     when the pick is a row with itself.
        // The body of the walk is straight-line (21 instructions per digit pair, 35 with a branch per block): what depends on the partner
        // only -- which blocks exist, which are mirrored, where the B-role key sits -- is folded into three counter addresses and two signs in
        // front of it (a block that does not exist counts into a spare vector of the wavefront that nobody reads), the consumed digits (one
        // per column, hardly ever more) are the OUTER loop.  Batch 29.6 -> 29.25 us per step, one chain 19.8 -> 19.65 (MI355X).
        const bool hasA = sA != SLOT_NONE, hasB = same ? hasA : sB != SLOT_NONE;
        const bool mirA = A < pr, mirB = same ? mirA : B < pr;
        const int Wk4 = 4 * (2 * nb - 1), sgnA4 = mirA ? -4 : 4, sgnB4 = mirB ? -4 : 4, tau = pk_sub ? -1 : 1;  // (byte offsets into the counters)
        unsigned char *const spare = reinterpret_cast<unsigned char *>(s_cnt + (size_t)QN * 3 * Kpad);
        unsigned char *const pN = reinterpret_cast<unsigned char *>(cN + (nb - 1));
        unsigned char *const pA = (hasA ? reinterpret_cast<unsigned char *>(dA) : spare) + 4 * (nb - 1);
        unsigned char *const pB = (hasB ? reinterpret_cast<unsigned char *>(same ? dA : dB) : spare) + 4 * (nb - 1) + (pk_sub ? Wk4 : 0) + sgnB4 * pk_shift;
        auto bump = [](unsigned char *at) { atomicAdd(reinterpret_cast<uint32_t *>(at), 1u); };
        auto pairs_of = [&](const Entry en) {
            using O = CellOps<Cell>;
            const Cell x = F::cell(en);
            const int at = x ? (int)s_cmap[F::col(en)] : 0;
            if (!at) return;  // empty cell, or a column that was not substituted
            const Cell ma = s_mA[at - 1];
            const uint32_t xm = O::minus(x), hm = O::minus(ma), la0 = O::plus(x) | xm;
            uint32_t h = O::plus(ma) | hm;
            while (h) {
                const int q_ = ctz32(h);
                h &= h - 1;
                const uint32_t xq = (hm >> q_) & 1u ? ~xm : xm;  // bit p: the signs of the two digits differ
                uint32_t la = la0;
                while (la) {
                    const int p_ = ctz32(la);
                    la &= la - 1;
                    int sgm = -(int)((xq >> p_) & 1u);
                    pin_vgpr(sgm);  // (kept as a mask: the compiler otherwise turns the AND below into compare + move + select)
                    const int d = q_ - p_, off = sgm & Wk4;  // key (d, sg) -> index sg (2 nb - 1) + d + nb - 1
                    bump(pN + off + 4 * d);
                    bump(pA + off + __mul24(d, sgnA4));
                    bump(pB + __mul24(off, tau) + __mul24(d, sgnB4));
                }
            }
        };
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (__any(u * QG < cnt)) pairs_of(e[u]);  // e[u] is empty beyond the list
        for (int j = l + CH * QG; __any(j < cnt); j += QG)
            if (j < cnt) pairs_of(rl[(size_t)off + (dense ? s_col[j] : j)]);
        lds_fence();
        int fnew = 0;  // a block (partner, new row) is created when one of its counts reached 2
        for (int k = l; k < K; k += QG) fnew |= cN[k] >= 2u;
        const bool gnew = valid && (((uint32_t)(__ballot(fnew != 0) >> qsh) & QMASK) != 0);
        UPD_TIMER_MARK(2)  // pair enumeration
        // ---- re-evaluate both blocks at once: two counts per lane and word, reduction inside the eight lanes of a block, their last lane publishes
        {
            const bool has = sX >= 0;
            const int ov = hdX.x;
            const float dl = __int_as_float(hdX.y);
            const uint32_t *d = half ? dB : dA;
            unsigned long long best = 0;
            int alive = 0;
#pragma unroll
            for (int u = 0; u < HW; ++u) {
                const int j = hl + u * HL;
                if (has && j < KW) {
                    const uint32_t o0 = wX[u] & 0xFFFFu, o1 = wX[u] >> 16;
                    const uint32_t n0 = o0 - d[2 * j], n1 = o1 - d[2 * j + 1];
                    if (n0 != o0 || n1 != o1) reinterpret_cast<DA_GLOBAL uint32_t *>(blk_ptr(c, sX) + 16)[j] = (n0 & 0xFFFFu) | (n1 << 16);
                    alive |= (n0 >= 2u) | (n1 >= 2u);
                    const uint32_t r0 = entry_rank(n0, ov, dl, c.method), r1 = entry_rank(n1, ov, dl, c.method);
                    const unsigned long long c0 = r0 ? (((unsigned long long)r0 << 8) | (unsigned)(2 * j)) : 0ull;
                    const unsigned long long c1 = r1 ? (((unsigned long long)r1 << 8) | (unsigned)(2 * j + 1)) : 0ull;
                    best = c0 > best ? c0 : best;
                    best = c1 > best ? c1 : best;
                }
            }
            best = part_row_max_u64<HL>(best);  // all lanes take part (the DPP source lanes must be active); every lane of a block's HL holds its result
            const bool any_alive = (((uint32_t)(__ballot(alive != 0) >> (qsh + HL * half)) & HMASK) != 0);
            if (has && hl == HL - 1) block_commit(c, sX, keyX, BlkHdr{ov, dl, (uint32_t)hdX.z, (uint32_t)hdX.w}, best, any_alive, false);
            if constexpr (STATS) deletes += (unsigned)__popcll(__ballot(has && hl == HL - 1 && !any_alive));  // (wave-uniform: the blocks this pass deleted, tallied once per workgroup)
        }
        UPD_TIMER_MARK(3)  // block updates
        // ---- block creation, ALL creating groups of the wavefront at once (a group = one new block (partner, new row)): the first bucket of the key
        // is probed by the group's lanes, its first free slot claimed by compare-and-swap (lost to another claimant: the next free one), the counts
        // come from the group's LDS counters two per lane, the best key by a reduction inside the DPP row.  One instruction stream for up to QN
        // creations instead of one wave-wide table_insert after the other: the launch lasts as long as its slowest wave, and in the first thousands
        // of steps that is a wave with four creations in a pass (round 6: batch 30.4 -> 29.5 us per step; steps 0 - 2000: update 35.1 -> 28.7 us).
        // A group whose first bucket has no free slot takes the wave-wide path below.
        bool gslow = false;  // this group's creation is left to table_insert
        if (__ballot(gnew)) {
            const unsigned long long keyN = pack_pair(pr, Nw);  // (the partner is older than the new row)
            const uint32_t baseN = (hash_pair(pr, Nw) & ~(BUCKET - 1)) & c.cmask;
            unsigned long long kN[SPL];
#pragma unroll
            for (int t = 0; t < SPL; ++t) kN[t] = gnew ? c.hkey[baseN + l * SPL + t] : 0ull;
            const RowInfo rp = load_row(c.rows, gnew ? pr : 0u);
            int pick_t = -1;  // a free slot of this lane: EMPTY, or a tombstone of an earlier launch
#pragma unroll
            for (int t = SPL - 1; t >= 0; --t)
                if (gnew && (kN[t] == KEY_EMPTY || (kN[t] >= KEY_TOMB_LO && kN[t] != c.tomb))) pick_t = t;
            uint32_t av = (uint32_t)(__ballot(pick_t >= 0) >> qsh) & QMASK;
            int nslot = -1;
            bool trying = gnew;
            while (__ballot(trying)) {
                const int cand = av ? ctz32(av) : -1;
                int ok = 0;
                if (trying && l == cand) {
                    const unsigned long long was = pick_t == 0 ? kN[0] : kN[SPL - 1];
                    ok = atomicCAS(gen(&c.hkey[baseN + l * SPL + pick_t]), was, keyN) == was;
                }
                const bool won = (((uint32_t)(__ballot(ok != 0) >> qsh)) & QMASK) != 0;
                int tsel = 0;
                if constexpr (SPL > 1) tsel = __shfl(pick_t, cand >= 0 ? qsh + cand : lane);
                if (trying) {
                    if (cand < 0) {
                        trying = false;
                        gslow = true;
                    } else if (won) {
                        nslot = (int)(baseN + (uint32_t)(cand * SPL + tsel));
                        trying = false;
                    } else
                        av &= av - 1;
                }
            }
            const bool made = nslot >= 0;
            const int ovn = n_overlap(rp, rnew);
            const float dln = fabsf(rp.lat - rnew.lat);
            unsigned long long bestn = 0;
            constexpr int CWN = ((sizeof(Cell) == 4 ? 24 : 60) + QG - 1) / QG;  // count words per lane
#pragma unroll
            for (int u = 0; u < CWN; ++u) {
                const int j = l + u * QG;
                if (made && j < KW) {
                    const uint32_t n0 = cN[2 * j], n1 = cN[2 * j + 1];  // (zero beyond K: nothing ever counts there)
                    if ((n0 | n1) > 65535u) c.g->error = E_COUNT_OVERFLOW;
                    reinterpret_cast<DA_GLOBAL uint32_t *>(blk_ptr(c, nslot) + 16)[j] = (n0 & 0xFFFFu) | (n1 << 16);
                    const uint32_t r0 = entry_rank(n0, ovn, dln, c.method), r1 = entry_rank(n1, ovn, dln, c.method);
                    const unsigned long long c0 = r0 ? (((unsigned long long)r0 << 8) | (unsigned)(2 * j)) : 0ull;
                    const unsigned long long c1 = r1 ? (((unsigned long long)r1 << 8) | (unsigned)(2 * j + 1)) : 0ull;
                    bestn = c0 > bestn ? c0 : bestn;
                    bestn = c1 > bestn ? c1 : bestn;
                }
            }
            bestn = part_row_max_u64<8>(bestn);  // (eight lanes, then the two halves of the group)
            if constexpr (QG == 16) bestn = dpp_max_u64<0x140, 0xF>(bestn);  // row_mirror: lane i <-> 15 - i
            if (made && l == QG - 1) {
                const uint32_t rank = (uint32_t)(bestn >> 8), bidx = (uint32_t)(bestn & 0xFF);
                store_hdr(c, nslot, ovn, dln, rank, bidx);
                c.hrank[nslot] = rank;
                hidx_ptr(c)[nslot] = (uint8_t)bidx;
                if (rank) {
                    group_note(c, nslot, 0ull, bound_word(rank, tie_word(pr, Nw, (int)bidx)));
                    fold_entry(c, rank, tie_word(pr, Nw, (int)bidx));
                }
            }
            if constexpr (STATS) inserts += (unsigned)__popcll(__ballot(made && l == 0));
        }
        // ---- rare: blocks beyond their first bucket (look-up, or creation in a full bucket) -- the whole wave, one group at a time
        const unsigned long long rare = __ballot(valid && (sA == SLOT_SLOW || sB == SLOT_SLOW || gslow));
        if constexpr (STATS) found += (unsigned)__popcll(__ballot(l == 0 && sA >= 0)) + (unsigned)__popcll(__ballot(l == 0 && sB >= 0));
        if (rare) {
#pragma unroll 1
            for (int qq = 0; qq < QN; ++qq) {  // a plain loop: the wave-wide table functions are emitted once
                if (!((rare >> (qq * QG)) & 1ull)) continue;
                const uint32_t rpr = (uint32_t)__builtin_amdgcn_readlane((int)pr, qq * QG);
                const int rsA = __builtin_amdgcn_readlane(sA, qq * QG), rsB = __builtin_amdgcn_readlane(sB, qq * QG);
                const int rnewb = __builtin_amdgcn_readlane((int)gslow, qq * QG);
                const uint32_t *rdA = s_cnt + (size_t)qq * 3 * Kpad, *rdB = rdA + Kpad, *rcN = rdB + Kpad;
                if (rsA == SLOT_SLOW) {
                    const uint32_t lo = min(A, rpr), hi = max(A, rpr);
                    const int slot = table_find_from(c, pack_pair(lo, hi), hash_pair(lo, hi), 1);
                    if (slot >= 0) {
                        int gone = 0;
                        table_update(c, slot, pack_pair(lo, hi), [&](int k, uint32_t old) { return old - rdA[k]; }, false, &gone);
                        if constexpr (STATS) deletes += (unsigned)gone, ++found;
                    }
                }
                if (rsB == SLOT_SLOW) {
                    const uint32_t lo = min(B, rpr), hi = max(B, rpr);
                    const int slot = table_find_from(c, pack_pair(lo, hi), hash_pair(lo, hi), 1);
                    if (slot >= 0) {
                        int gone = 0;
                        table_update(c, slot, pack_pair(lo, hi), [&](int k, uint32_t old) { return old - rdB[k]; }, false, &gone);
                        if constexpr (STATS) deletes += (unsigned)gone, ++found;
                    }
                }
                if (rnewb) {
                    table_insert(c, rpr, Nw, load_row(c.rows, rpr), rnew, [&](int k) { return rcN[k]; }, nullptr, false);
                    if constexpr (STATS) ++inserts;
                }
            }
        }
        UPD_TIMER_PASS_END  // slow path + block creation
        lds_fence();  // the next pass overwrites the counters
    }
    UPD_TIMER_FLUSH
}

// special_pairs: the pairs among the rows a step modified -- (A,A), (A,B), (B,B): their blocks are re-counted from scratch; (A,N), (B,N),
// (N,N): new -- from the exact counts the substitution block left in sp_cnt (select_body used to write them itself: the table is now
// written by this kernel only).  One wavefront per pair.  Like every block this kernel re-evaluates they pass their best entry to fold_entry,
// changed or not.
// Two workgroups per chain share the work, one pair per wavefront and ONE turn each (find -> re-count / create: the longest dependent chain
// of this kernel; with all six in one workgroup of four waves it took two turns): part 0 the four pairs with A, part 1 (B,N), (N,N) and the
// listed entries.
template <class Cell> __device__ __forceinline__ void special_pairs(ChainDev *gq, const UpdStep<Cell> &u, int part) {
    const Ctx &c = u.c;
    const uint32_t A = u.A, B = u.B, Nw = u.Nw;
    const bool same = A == B;
    const DA_GLOBAL uint32_t *spc = (const DA_GLOBAL uint32_t *)gq->sp_cnt;
    const RowInfo ra = load_row(c.rows, A), rb = load_row(c.rows, B), rn = load_row(c.rows, Nw);
    const int lane = lane_id();
    for (int sp = part * UPD_WAVES + wave_id(); sp < min(6, (part + 1) * UPD_WAVES); sp += UPD_WAVES) {
        uint32_t lo = A, hi = A;
        bool active = true, existed = false;
        int row = 0;  // row of the count vectors left by the substitution block: (A,A) (A,B) (B,B) (A,N) (B,N) (N,N)
        switch (sp) {
        case 0: lo = A, hi = A, existed = true, row = 0; break;
        case 1: lo = A, hi = B, existed = true, active = !same, row = 1; break;
        case 2: lo = A, hi = Nw, row = 3; break;
        case 3: lo = B, hi = B, existed = true, active = !same, row = 2; break;
        case 4: lo = B, hi = Nw, active = !same, row = 4; break;
        default: lo = Nw, hi = Nw, row = 5; break;
        }
        if (!active) continue;
        const DA_GLOBAL uint32_t *cnt = spc + (size_t)row * c.Kpad;
        const unsigned long long key = pack_pair(lo, hi);
        const int slot = existed ? table_find(c, key, hash_pair(lo, hi)) : -1;
        unsigned long long w = 0;
        if (slot >= 0)
            w = table_update(c, slot, key, [&](int k, uint32_t) { return cnt[k]; });
        else {
            int f = 0;
            for (int k = lane; k < c.K; k += WAVE) f |= cnt[k] >= 2u;
            if (__any(f)) {
                const RowInfo xa = pick_row(lo == Nw, rn, pick_row(lo == A, ra, rb)), xb = pick_row(hi == Nw, rn, pick_row(hi == A, ra, rb));
                table_insert(c, lo, hi, xa, xb, [&](int k) { return cnt[k]; }, &w);
            }
        }
        (void)w;  // (both paths have passed the block's best entry to fold_entry)
    }
    // the entries the search listed (they touch exactly one of A / B and reach the untouched entry): a block whose other row is a partner
    // row is re-evaluated by the partner waves of this launch, which pass its best entry on themselves; the others are unchanged -- passed on here.
    // Partner row = a row with digits in a substituted column: bit r of the row bitmaps of those columns (A, B, N are never the other row)
    if (c.rword && part == 1) {
        const int nl = (int)gq->l_n[(gq->iter - 1) & 1];
        const CandEntry *ll = &gq->l_list[(gq->iter - 1) & 1][0];
        const uint32_t *colbits = gq->colbits;
        const int cbw = gq->cb_words;
        for (int e = (wave_id() + 2) % UPD_WAVES; e < nl; e += UPD_WAVES) {  // (waves 2 and 3 first: 0 and 1 had a pair)
            const unsigned long long tw = ll[e].tie;
            const uint32_t i0 = (uint32_t)((tw >> 7) & 0xFFFFFFu), i1 = (uint32_t)(tw >> 31);
            const uint32_t r = (i0 == A || i0 == B) ? i1 : i0;
            int hit = 0;
            for (int k = lane; k < u.m; k += WAVE) hit |= (int)((colbits[(size_t)u.mcol[k] * cbw + (r >> 5)] >> (r & 31)) & 1u);
            if (!__any(hit) && lane == 0) fold_entry(c, (uint32_t)ll[e].rank, tw);
        }
    }
}

// update_body: the partner rows [block_y * NWV * QN + ..., stride grid_y * NWV * QN) of chain `gq`'s current step, by a
// workgroup of NWV wavefronts (k_iter_update: 256-thread blocks, grid = chains x blocks per chain).
template <class Cell, int NWV, bool STATS>
__device__ __forceinline__ void update_body(ChainDev *gq, bool in_range, int block_y, int grid_y) {
    constexpr int NTHR = NWV * WAVE;
    // Grid (chains padded to a multiple of 8, blocks per chain): the chain index is the FAST grid dimension.  Workgroups go to
    // the XCDs round-robin by their linear id, so all blocks of chain c -- and block c of k_iter_select, which has the same
    // linear id modulo 8 -- run on XCD c mod 8: the table lines, bounds and lists of a chain stay in ONE of the eight
    // non-coherent L2s instead of being spread over all of them.
    const UpdStep<Cell> u = load_upd_step<Cell>(gq);
    if (!in_range || u.done) return;
    if (block_y >= grid_y - 2) {  // the last two blocks of a chain: the six blocks of the pairs among {A, B, new row}
        special_pairs<Cell>(gq, u, block_y - (grid_y - 2));
        return;
    }
    grid_y -= 2;
    // the grid is sized for the partner counts of the first steps of a chain (thousands); later most blocks have nothing
    // to do and leave before the hand-off is copied
    if (block_y * (NWV * QN) >= u.n_partners) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
    UpdLds<Cell> s;
    s.carve(smem, smem + align16(UpdLds<Cell>::table_bytes(u.c.n_out)) + (size_t)wid * UpdLds<Cell>::wave_bytes(u.c.Kpad), u.c.n_out);
    const int total_waves = grid_y * NWV, gw = block_y * NWV + wid;
    // ---- ONE vector round trip: the group's first partner reference and the new row's record leave together with the
    // hand-off of k_iter_select (they used to wait behind the hand-off barrier: two more dependent round trips)
    const unsigned long long ref0 = gw * QN + (lane >> QG_LOG2) < u.n_partners ? u.plist[gw * QN + (lane >> QG_LOG2)] : 0ull;
    const RowInfo rnew = load_row(u.c.rows, u.Nw);
    __shared__ unsigned int s_stat[3];
    if constexpr (STATS)
        if (tid < 3) s_stat[tid] = 0;
    copy_handoff<Cell>(u, s, tid, NTHR);  // one pass, one barrier (the column map arrives ready-made)
    __syncthreads();
    unsigned int found = 0, inserts = 0, deletes = 0;
    update_partners<Cell, STATS>(gq, u, s, gw * QN, total_waves * QN, u.n_partners, ref0, rnew, found, inserts, deletes);
    if constexpr (!STATS) return;
    // statistics: summed per block in LDS, then ONE pair of device atomics per block.  (Four atomics per wave on one line
    // of the chain descriptor -- 640 per chain and launch, from all XCDs -- serialise at ~12 ns each and every launch had
    // to wait for them; the partner / cell counts are added by k_iter_select, which knows them without counting.)
    if (lane == 0 && (found | inserts | deletes)) {
        if (found) atomicAdd(&s_stat[0], found);
        if (inserts) atomicAdd(&s_stat[1], inserts);
        if (deletes) atomicAdd(&s_stat[2], deletes);
    }
    __syncthreads();
    if (tid == 0) {
        if (s_stat[0]) atomicAdd(&gq->st_found, (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&gq->st_inserts, (unsigned long long)s_stat[1]);
        if (s_stat[1] != s_stat[2]) atomicAdd(&gq->n_live, s_stat[1] - s_stat[2]);  // blocks created less blocks deleted by this workgroup (modulo 2^32)
    }
}
template <class Cell, bool STATS>
__global__ void __launch_bounds__(UPD_THREADS, DA_UPD_OCC) __attribute__((amdgpu_num_sgpr(DA_UPD_SGPRS))) k_iter_update(ChainDev *chains, int n_chains) {
    // Grid (chains padded to a multiple of 8, blocks per chain): the chain index is the FAST grid dimension (see update_body)
#ifdef DA_STEP_CLOCKS
    ChainDev *gk = &chains[(int)blockIdx.x < n_chains ? blockIdx.x : 0];
    const bool clk_on = (int)blockIdx.x < n_chains && !gk->done;
    const int clk_par = (gk->iter - 1) & 1;
    if (clk_on && threadIdx.x == 0) CLK_MARK(gk, clk_par, 2, true);
#endif
    update_body<Cell, UPD_WAVES, STATS>(&chains[(int)blockIdx.x < n_chains ? blockIdx.x : 0], (int)blockIdx.x < n_chains, (int)blockIdx.y, (int)gridDim.y);  // clamped: the descriptor read is unconditional
#ifdef DA_STEP_CLOCKS
    __syncthreads();
    if (clk_on && threadIdx.x == 0) CLK_MARK(gk, clk_par, 3, false);
#endif
}

// ================================================================================= column-sharded chains (cmvm_shard.h)
// The chain holds the digits of a slice of the columns and a replica of the pair table; counts are sums over columns,
// exchanged between the ranks as int32 slabs (all-reduce(sum) between the kernels below, driven by cmvm_shard.cc).

// grid (ceil(n_pairs / 4)): one wave per pair of input rows: partial counts over the own columns -> cs_init [n_pairs][K]
template <class Cell> __global__ void __launch_bounds__(256) k_cs_init_counts(ChainDev *g) {
    const Ctx c = make_ctx(g, 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem) + (size_t)wave_id() * c.Kpad;
    const long long n_in = g->n_in, n_pairs = n_in * (n_in + 1) / 2;
    const long long p = (long long)blockIdx.x * (blockDim.x / WAVE) + wave_id();
    if (p >= n_pairs) return;
    long long i1 = (long long)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (i1 * (i1 + 1) / 2 > p) --i1;
    while ((i1 + 1) * (i1 + 2) / 2 <= p) ++i1;
    const uint32_t hi = (uint32_t)i1, lo = (uint32_t)(p - i1 * (i1 + 1) / 2);
    count_row_pair<Cell>(c, reinterpret_cast<const typename RowFmt<Cell>::Entry *>(g->rlist), lo, hi, cnt);
    for (int k = lane_id(); k < c.K; k += WAVE) g->cs_init[(size_t)p * c.K + k] = (int32_t)cnt[k];
}
// same grid: the summed counts into the (replicated) table
template <class Cell> __global__ void __launch_bounds__(256) k_cs_init_table(ChainDev *g) {
    const Ctx c = make_ctx(g, 0);
    const long long n_in = g->n_in, n_pairs = n_in * (n_in + 1) / 2;
    const long long p = (long long)blockIdx.x * (blockDim.x / WAVE) + wave_id();
    if (p >= n_pairs) return;
    long long i1 = (long long)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (i1 * (i1 + 1) / 2 > p) --i1;
    while ((i1 + 1) * (i1 + 2) / 2 <= p) ++i1;
    const uint32_t hi = (uint32_t)i1, lo = (uint32_t)(p - i1 * (i1 + 1) / 2);
    const int32_t *cnt = g->cs_init + (size_t)p * c.K;
    int f = 0;
    for (int k = lane_id(); k < c.K; k += WAVE) f |= cnt[k] >= 2;
    if (!__any(f)) return;
    if (c.method < 0) {  // unknown method string with a non-empty table: the reference throws here
        if (lane_id() == 0) g->unknown_hit = 1;
        return;
    }
    table_insert(c, lo, hi, load_row(c.rows, lo), load_row(c.rows, hi), [&](int k) { return (uint32_t)cnt[k]; });
}

// one block: union of the partner rows of all ranks (summed flag fields != 0), ascending row ids -> cs_uni, cs_nuni
// `report` (pinned host memory, mapped): {sequence number of the step, size of the union, the three summed status words} -- what the host needs
// between the two exchanges of a step, written by the device itself: no device-to-host copies and no stream synchronisation per step
// (each of the two small copies was a DMA operation of ~10 us; round 6).  The sequence number is stored last, behind a system-scope fence.
__global__ void __launch_bounds__(1024) k_cs_union(ChainDev *g, volatile int *report, int seq, int flag_words) {
    __shared__ int s_part[16], s_base[17];
    const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
    const int n_rows = (int)g->Nw;  // rows that existed before this step's new row
    const int per = (n_rows + 1023) / 1024;
    const int r0 = min(tid * per, n_rows), r1 = min(r0 + per, n_rows);
    int mine = 0;
    for (int r = r0; r < r1; ++r) mine += ((g->cs_flags[r >> 2] >> (8 * (r & 3))) & 0xFF) != 0;
    int inc = mine;
    for (int o = 1; o < WAVE; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == WAVE - 1) s_part[wid] = inc;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < 16; ++w) {
            s_base[w] = run;
            run += s_part[w];
        }
        s_base[16] = run;
        g->cs_nuni = run;
        if (report) {
            report[1] = run;
            report[2] = g->cs_flags[flag_words];
            report[3] = g->cs_flags[flag_words + 1];
            report[4] = g->cs_flags[flag_words + 2];
            __threadfence_system();
            report[0] = seq;
        }
    }
    __syncthreads();
    int at = s_base[wid] + inc - mine;
    for (int r = r0; r < r1; ++r)
        if (((g->cs_flags[r >> 2] >> (8 * (r & 3))) & 0xFF) != 0) g->cs_uni[at++] = (uint32_t)r;
}

// grid (ceil(n_union / 4)): one wave per row of the union: this rank's partial count changes over its own substituted
// columns -> slab [6 + 3 u .. 6 + 3 u + 2][K] = {lost with A, lost with B, gained with the new row}
template <class Cell> __global__ void __launch_bounds__(256) k_cs_partial(ChainDev *g) {
    using F = RowFmt<Cell>;
    using Entry = typename F::Entry;
    const Ctx c = make_ctx(g, 0);
    const int m = g->m, nb = c.n_bits, n_out = c.n_out, K = c.K, Kpad = c.Kpad;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Cell *s_mA = reinterpret_cast<Cell *>(smem);
    Cell *s_mB = s_mA + n_out;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_mB + n_out);  // [4][3][Kpad]
    uint16_t *s_cmap = reinterpret_cast<uint16_t *>(s_cnt + 4 * 3 * Kpad);
    const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
    for (int j = tid; j < n_out; j += 256) s_cmap[j] = 0;
    __syncthreads();
    for (int j = tid; j < m; j += 256) {
        s_cmap[g->mcol[j]] = (uint16_t)(j + 1);
        s_mA[j] = reinterpret_cast<const Cell *>(g->mA)[j];
        s_mB[j] = reinterpret_cast<const Cell *>(g->mB)[j];
    }
    __syncthreads();
    const int u = (int)blockIdx.x * 4 + wid;
    if (u >= g->cs_nuni) return;
    const uint32_t A = g->A, B = g->B, r = g->cs_uni[u];
    const bool same = A == B;
    uint32_t *dA = s_cnt + (size_t)wid * 3 * Kpad, *dB = dA + Kpad, *cN = dB + Kpad;
    for (int k = lane; k < 3 * Kpad; k += WAVE) dA[k] = 0;
    lds_fence();
    const da_u2 ref = g->rowoff[r];
    const Entry *row = reinterpret_cast<const Entry *>(g->rlist) + ref.x;
    for (int j = lane; j < (int)ref.y; j += WAVE) {
        const Entry en = row[j];
        const Cell x = F::cell(en);
        if (!x) continue;
        const int at = (int)s_cmap[F::col(en)];
        if (!at) continue;
        const Cell ma = s_mA[at - 1], mb = s_mB[at - 1];
        for_pairs_part<Cell>(ma, x, A < r, nb, [&](int k) { atomicAdd(&dA[k], 1u); });
        if (same)
            for_pairs_part<Cell>(mb, x, A < r, nb, [&](int k) { atomicAdd(&dA[k], 1u); });
        else
            for_pairs_part<Cell>(mb, x, B < r, nb, [&](int k) { atomicAdd(&dB[k], 1u); });
        for_pairs_cross<Cell>(x, ma, nb, [&](int k) { atomicAdd(&cN[k], 1u); });
    }
    lds_fence();
    int32_t *out = g->cs_slab + (size_t)(6 + 3 * u) * K;
    for (int k = lane; k < K; k += WAVE) {
        out[k] = (int32_t)dA[k];
        out[K + k] = (int32_t)dB[k];
        out[2 * K + k] = (int32_t)cN[k];
    }
}

// grid (ceil((n_union + 6) / 4)): the summed slab into the table.  Waves 0-5: the six pairs among {A, B, new} (their
// blocks are replaced); wave 6 + u: row u of the union (blocks with A and B reduced, block with the new row created).
template <class Cell> __global__ void __launch_bounds__(256) k_cs_apply(ChainDev *g) {
    Ctx c = make_ctx(g, 2 * g->iter - 1);
    // as k_iter_update (load_upd_step): the step being applied is iter - 1; the best entry of every block this launch writes or re-evaluates that
    // reaches the entry the step leaves untouched (spec) is listed for the next pick (fold_entry)
    c.rword = g->spec[(g->iter - 1) & 1].word;
    c.cn = &g->c_n[(g->iter - 1) & 1];
    c.cl = &g->c_list[(g->iter - 1) & 1][0];
    const int K = c.K;
    const int w = (int)blockIdx.x * 4 + wave_id();
    const uint32_t A = g->A, B = g->B, Nw = g->Nw;
    const bool same = A == B;
    if (w == 5 && c.rword) {
        // the entries the search listed (they touch exactly one of A / B and reach the untouched entry): a block whose other row is in the union of
        // partner rows is re-evaluated by that row's wave, which passes its best entry on itself; the others are unchanged -- passed on here
        // (special_pairs does the same for ordinary chains).  In the union <=> the row's summed flag field is not zero.
        const int nl = (int)g->l_n[(g->iter - 1) & 1];
        const CandEntry *ll = &g->l_list[(g->iter - 1) & 1][0];
        if (lane_id() == 0)
            for (int e = 0; e < nl; ++e) {
                const unsigned long long tw = ll[e].tie;
                const uint32_t i0 = (uint32_t)((tw >> 7) & 0xFFFFFFu), i1 = (uint32_t)(tw >> 31);
                const uint32_t r = (i0 == A || i0 == B) ? i1 : i0;
                if (((g->cs_flags[r >> 2] >> (8 * (r & 3))) & 0xFF) == 0) fold_entry(c, (uint32_t)ll[e].rank, tw);
            }
    }
    if (w < 6) {
        uint32_t lo = A, hi = A;
        bool active = true, existed = false;
        switch (w) {
        case 0: lo = A, hi = A, existed = true; break;
        case 1: lo = A, hi = B, existed = true, active = !same; break;
        case 2: lo = B, hi = B, existed = true, active = !same; break;
        case 3: lo = A, hi = Nw; break;
        case 4: lo = B, hi = Nw, active = !same; break;
        default: lo = Nw, hi = Nw; break;
        }
        if (!active) return;
        const int32_t *cnt = g->cs_slab + (size_t)w * K;
        const unsigned long long key = pack_pair(lo, hi);
        const int slot = existed ? table_find(c, key, hash_pair(lo, hi)) : -1;
        if (slot >= 0)
            table_update(c, slot, key, [&](int k, uint32_t) { return (uint32_t)cnt[k]; });
        else {
            int f = 0;
            for (int k = lane_id(); k < K; k += WAVE) f |= cnt[k] >= 2;
            if (__any(f)) table_insert(c, lo, hi, load_row(c.rows, lo), load_row(c.rows, hi), [&](int k) { return (uint32_t)cnt[k]; });
        }
        return;
    }
    const int u = w - 6;
    if (u >= g->cs_nuni) return;
    const uint32_t r = g->cs_uni[u];
    const int32_t *dA = g->cs_slab + (size_t)(6 + 3 * u) * K, *dB = dA + K, *cN = dB + K;
    {
        const uint32_t lo = min(A, r), hi = max(A, r);
        const int slot = table_find(c, pack_pair(lo, hi), hash_pair(lo, hi));
        if (slot >= 0) table_update(c, slot, pack_pair(lo, hi), [&](int k, uint32_t old) { return old - (uint32_t)dA[k]; });
    }
    if (!same) {
        const uint32_t lo = min(B, r), hi = max(B, r);
        const int slot = table_find(c, pack_pair(lo, hi), hash_pair(lo, hi));
        if (slot >= 0) table_update(c, slot, pack_pair(lo, hi), [&](int k, uint32_t old) { return old - (uint32_t)dB[k]; });
    }
    int f = 0;
    for (int k = lane_id(); k < K; k += WAVE) f |= cN[k] >= 2;
    if (__any(f)) table_insert(c, r, Nw, load_row(c.rows, r), load_row(c.rows, Nw), [&](int k) { return (uint32_t)cN[k]; });
    if (lane_id() == 0) atomicAdd(&g->st_partners, 1ull);
}

// ------------------------------------------------------------------------------------------------ k_extract
// grid (ceil(n_out / 4), n_chains): one wave per column compacts the surviving (row, cell) entries in row order.
template <class Cell> __global__ void __launch_bounds__(256) k_extract(ChainDev *chains) {
    using O = CellOps<Cell>;
    using F = RowFmt<Cell>;
    ChainDev &ch = chains[blockIdx.y];
    int j = blockIdx.x * (blockDim.x / WAVE) + wave_id();
    if (j >= ch.n_out) return;
    const auto *rl = reinterpret_cast<const typename F::Entry *>(ch.rlist);
    int lane = lane_id(), len = ch.collen[j], out = 0;
    for (int base = 0; base < len; base += WAVE) {
        int e = base + lane;
        Cell c = 0;
        uint32_t r = 0;
        if (e < len) {
            const unsigned long long ref = ch.collist[(size_t)j * ch.lcap + e];
            r = ref_row(ref);
            const auto *row = rl + ref_off(ref);
            if ((int)r < ch.n_in)
                c = F::cell(row[j]);  // dense input row
            else {  // sparse row, sorted by column: binary search for column j (listed here, so it is present)
                int lo = 0, hi = (int)ref_len(ref) - 1;
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (F::col(row[mid]) < (uint32_t)j)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                const auto en = row[lo];
                c = F::col(en) == (uint32_t)j ? F::cell(en) : (Cell)0;
            }
        }
        unsigned long long nz = __ballot(c != 0);
        if (c != 0) {
            size_t at = (size_t)j * ch.lcap + out + __popcll(nz & ((1ull << lane) - 1));
            ch.fin_row[at] = r;
            ch.fin_cell[at] = (unsigned long long)O::plus(c) | ((unsigned long long)O::minus(c) << 32);
        }
        out += __popcll(nz);
    }
    if (lane == 0) ch.fin_count[j] = (uint32_t)out;
}

// ------------------------------------------------------------------------------------------------ k_pack
// grid (n_chains): prefix sum of the per-column counts and a dense copy of the surviving digits and of the row
// latencies, so that the download is proportional to the result instead of to the list capacities.
__global__ void __launch_bounds__(256) k_pack(ChainDev *chains) {
    ChainDev &ch = chains[blockIdx.x];
    __shared__ uint32_t s_total;
    const int n_out = ch.n_out, tid = threadIdx.x, lane = lane_id(), nw = blockDim.x / WAVE;
    if (tid < WAVE) {  // exclusive prefix by the first wave, 64 columns per step
        uint32_t run = 0;
        for (int base = 0; base < n_out; base += WAVE) {
            int j = base + lane;
            uint32_t v = j < n_out ? ch.fin_count[j] : 0u, inc = v;
            for (int o = 1; o < WAVE; o <<= 1) {
                uint32_t t = (uint32_t)__shfl_up((int)inc, o);
                if (lane >= o) inc += t;
            }
            if (j < n_out) ch.fin_start[j] = run + inc - v;
            run += (uint32_t)__shfl((int)inc, WAVE - 1);
        }
        if (lane == 0) {
            ch.fin_start[n_out] = run;
            ch.pk_total = run;
            s_total = run;
        }
    }
    __syncthreads();
    if (s_total > (uint32_t)ch.pk_cap) {
        if (tid == 0) ch.error = E_LIST_CAPACITY;
        return;
    }
    for (int j = wave_id(); j < n_out; j += nw) {
        const uint32_t cnt = ch.fin_count[j], dst = ch.fin_start[j];
        const size_t src = (size_t)j * ch.lcap;
        for (uint32_t e = lane; e < cnt; e += WAVE) {
            ch.pk_row[dst + e] = ch.fin_row[src + e];
            ch.pk_cell[dst + e] = ch.fin_cell[src + e];
        }
    }
    for (int r = tid; r < ch.n_rows; r += blockDim.x) ch.pk_lat[r] = ch.rows[r].lat;
}

// ------------------------------------------------------------------------------------------------ k_gather
// grid (blocks, pieces): the result arrays of all chains of a batch (seven per chain, scattered over the arena) copied into ONE
// contiguous buffer, which then leaves in a single device-to-host copy (448 small copies per 64-chain batch before: each a DMA
// operation of its own, ~10 us apiece on one stream).  Sources start on 256-byte boundaries, destinations on 64-byte ones.
struct GatherPiece {
    const void *src;
    unsigned long long dst_off, bytes;
};
__global__ void __launch_bounds__(256) k_gather(const GatherPiece *pieces, unsigned char *dst) {
    const GatherPiece pc = pieces[blockIdx.y];
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const size_t n16 = (size_t)pc.bytes / 16;
    const da_i4 *s16 = reinterpret_cast<const da_i4 *>(pc.src);
    da_i4 *d16 = reinterpret_cast<da_i4 *>(dst + pc.dst_off);
    for (size_t i = t0; i < n16; i += stride) d16[i] = s16[i];
    const unsigned char *s1 = reinterpret_cast<const unsigned char *>(pc.src);
    unsigned char *d1 = dst + pc.dst_off;
    for (size_t i = n16 * 16 + t0; i < (size_t)pc.bytes; i += stride) d1[i] = s1[i];
}

// ------------------------------------------------------------------------------------------------ k_col_dist
// Stage-1 distance matrix: d0[a][b] = sum_i nnzNAF(M[i,a] - M[i,b]), d1 with '+'.  grid (ceil(W/16), ceil(W/16)),
// block 16x16: a 16x16 tile of (a,b); the two 16-column strips of M are staged through LDS 64 rows at a time.
__global__ void __launch_bounds__(256) k_col_dist(const int32_t *aug, int n_in, int W, long long *d0, long long *d1) {
    __shared__ int32_t sa[64][17], sb[64][17];
    int ta = threadIdx.y, tb = threadIdx.x;
    int a = blockIdx.y * 16 + ta, b = blockIdx.x * 16 + tb;
    int tid = ta * 16 + tb;
    long long acc0 = 0, acc1 = 0;
    for (int base = 0; base < n_in; base += 64) {
        for (int q = tid; q < 64 * 16; q += 256) {
            int i = base + q / 16, c = q % 16;
            int ca = blockIdx.y * 16 + c, cb = blockIdx.x * 16 + c;
            sa[q / 16][c] = (i < n_in && ca < W) ? aug[(size_t)i * W + ca] : 0;
            sb[q / 16][c] = (i < n_in && cb < W) ? aug[(size_t)i * W + cb] : 0;
        }
        __syncthreads();
        int rows = min(64, n_in - base);
        for (int i = 0; i < rows; ++i) {
            int32_t x = sa[i][ta], y = sb[i][tb];
            acc0 += naf_weight(x - y);
            acc1 += naf_weight(x + y);
        }
        __syncthreads();
    }
    if (a < W && b < W) {
        d0[(size_t)a * W + b] = acc0;
        d1[(size_t)a * W + b] = acc1;
    }
}

// NAF digits of a flat int32 array (int_arr_to_csd, bit_decompose.cc:22-42): out[i][b] in {-1,0,1}
__global__ void __launch_bounds__(256) k_naf_digits(const int32_t *x, long long n, int N, int8_t *out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p, m;
    naf_masks(x[i], p, m);
    for (int b = 0; b < N; ++b) out[i * N + b] = (int8_t)(((p >> b) & 1) - ((m >> b) & 1));
}
__global__ void __launch_bounds__(256) k_absmax(const int32_t *x, long long n, unsigned int *out) {
    unsigned int mx = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int32_t v = x[i];
        mx = max(mx, (unsigned int)(v < 0 ? -v : v));
    }
    mx = wave_max_u32(mx);
    if (lane_id() == 0) atomicMax(out, mx);
}

// =================================================================================================== host side

namespace {

struct DeviceBuffer {  // grow-only device allocation reused across calls
    void *ptr = nullptr;
    size_t cap = 0;
    void *get(size_t bytes) {
        if (bytes > cap) {
            if (ptr) (void)hipFree(ptr);
            ptr = nullptr;
            cap = 0;
            size_t want = bytes + bytes / 8;
            HIP_CHECK(hipMalloc(&ptr, want));
            cap = want;
        }
        return ptr;
    }
    ~DeviceBuffer() {
        if (ptr) (void)hipFree(ptr);
    }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// spin-wait step of the two launch threads' hand-shake: a pause for the first few thousand polls (the partner answers within microseconds while both are
// queueing launches), then the core is given up between polls -- the waits that last (the main thread waiting for the device, the helper between
// windows) must not hold a core at 100 % (ranks of one host share its cores)
inline void spin_wait_step(unsigned &polls) {
    if (++polls < 4096u) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield");
#endif
    } else if (polls < 8192u)
        std::this_thread::yield();
    else
        std::this_thread::sleep_for(std::chrono::microseconds(50));
}

// Events of one call, destroyed however the call ends (a HIP error or the termination guard of the greedy loop used to leak
// the timing, window and sample events of the call -- up to 12 k of them).
struct EventGuard {
    std::vector<hipEvent_t> all;
    hipEvent_t make(unsigned flags = 0) {
        hipEvent_t e = nullptr;
        HIP_CHECK(flags ? hipEventCreateWithFlags(&e, flags) : hipEventCreate(&e));
        all.push_back(e);
        return e;
    }
    ~EventGuard() {
        for (hipEvent_t e : all) (void)hipEventDestroy(e);
    }
};

struct Carver {  // bump allocator over the arena; first pass sizes, second pass assigns
    unsigned char *base;
    size_t off = 0;
    explicit Carver(unsigned char *b) : base(b) {}
    template <class T> T *take(size_t count) {
        off = align_up(off, 256);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

uint32_t pow2_ceil(uint64_t v) {
    uint32_t p = 1;
    while (p < v && p < (1u << 31)) p <<= 1;
    return p;
}

}  // namespace

struct PinnedBuffer {  // grow-only pinned host allocation reused across calls
    void *ptr = nullptr;
    size_t cap = 0;
    void *get(size_t bytes) {
        if (bytes > cap) {
            if (ptr) (void)hipHostFree(ptr);
            ptr = nullptr;
            cap = 0;
            size_t want = bytes + bytes / 4 + 4096;
            HIP_CHECK(hipHostMalloc(&ptr, want, hipHostMallocDefault));
            cap = want;
        }
        return ptr;
    }
    ~PinnedBuffer() {
        if (ptr) (void)hipHostFree(ptr);
    }
};

struct HipBackend::Impl {
    int device = 0;
    PinnedBuffer pinned, pinned_up;  // staging of the downloads / of the upload
    hipStream_t stream = nullptr;
    static constexpr int MAX_LANES = 8;
    hipStream_t lanes[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // greedy-loop streams of the chain groups
    int launch_threads = 2;  // host threads queueing the greedy loop's launches (each its share of the chain groups); DA4ML_HIP_LAUNCH_THREADS
    int n_lanes = 4;  // chain groups = greedy-loop streams.  Measured (C3 batch 64, loop ms): 2 -> 880, 3 -> 871, 4 -> 838, 5..8 -> 1470:
                      // four hardware queues; the poll stream's rare copies share one of them at no visible cost
    int upd_total_blocks = 2560;  // k_iter_update blocks over all chains of a batch (4 waves x 4 groups each); measured (C3 batch 64,
                                  // solves/s): 1024: 45.3, 1536: 53.3, 2048: 53.8, 2560: 55.5, 4096: 47.5
    DeviceBuffer arena, desc_buf, io_buf, gather_buf, piece_buf;  // gather_buf: the results of a batch, contiguous, before they leave
    unsigned int *d_done = nullptr;
    unsigned int *h_done = nullptr;  // pinned, two words: done counters of alternating poll windows
    hipStream_t poll_stream = nullptr;
    GpuTimings timings;
    double table_scale = 1.0;  // grows on E_TABLE_CAPACITY retries
};

HipBackend::HipBackend(int device) : impl_(new Impl) {
    impl_->device = device;
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&impl_->stream, hipStreamNonBlocking));
    impl_->lanes[0] = impl_->stream;  // the first group runs on the main stream (hardware queues are a scarce resource)
    for (int l = 1; l < Impl::MAX_LANES; ++l) HIP_CHECK(hipStreamCreateWithFlags(&impl_->lanes[l], hipStreamNonBlocking));
    if (const char *e = std::getenv("DA4ML_HIP_TABLE_SCALE")) impl_->table_scale = std::max(1e-4, std::atof(e));
    if (const char *e = std::getenv("DA4ML_HIP_ROW_SCALE")) row_scale_ = std::max(1e-4, std::atof(e));
    if (const char *e = std::getenv("DA4ML_HIP_UPD_BLOCKS")) impl_->upd_total_blocks = std::max(2, std::atoi(e));
    if (const char *e = std::getenv("DA4ML_HIP_LAUNCH_THREADS")) impl_->launch_threads = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("DA4ML_HIP_LANES")) impl_->n_lanes = std::max(1, std::min((int)Impl::MAX_LANES, std::atoi(e)));
    HIP_CHECK(hipMalloc(&impl_->d_done, sizeof(unsigned int)));
    HIP_CHECK(hipHostMalloc(&impl_->h_done, 2 * sizeof(unsigned int), hipHostMallocDefault));
    HIP_CHECK(hipStreamCreateWithFlags(&impl_->poll_stream, hipStreamNonBlocking));
    Log2Table t = measure_log2_table();
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_log2), &t, sizeof t));
}
HipBackend::~HipBackend() {
    (void)hipSetDevice(impl_->device);
    if (impl_->d_done) (void)hipFree(impl_->d_done);
    if (impl_->h_done) (void)hipHostFree(impl_->h_done);
    if (impl_->poll_stream) (void)hipStreamDestroy(impl_->poll_stream);
    if (impl_->stream) (void)hipStreamDestroy(impl_->stream);
    for (int l = 1; l < Impl::MAX_LANES; ++l)
        if (impl_->lanes[l]) (void)hipStreamDestroy(impl_->lanes[l]);
}
const GpuTimings &HipBackend::timings() const { return impl_->timings; }
void HipBackend::reset_timings() { impl_->timings = GpuTimings{}; }
void *HipBackend::stream() const { return impl_->stream; }

namespace {

struct Geometry {
    bool wide;  // 64-bit cells and 16-byte list entries (more than 12 digits or more than 256 columns)
    int n_mant = 0;  // distinct non-power-of-two step mantissas of the inputs (StepLog2): rows of the -log2f table
    int n_bits, K, Kpad, rcap, lcap, gs_log2, n_groups, pk_cap, pb_log2;
    uint32_t C, rl_cap;
};

// Dynamic LDS of a k_iter_select2 block WITHOUT the optional claim area (pick_body's carve): B's list, six count vectors, five per-column arrays
size_t sel2_fixed_lds(const ChainJob &job, const Geometry &g) {
    const size_t no = (size_t)job.n_out, entb = g.wide ? 16 : 4;
    return no * entb + 6 * (size_t)g.Kpad * 4 + (5 * no + 1) * 4;
}
// What the device leaves for it: the per-workgroup LDS limit less the kernel's STATIC __shared__ arrays (the search block's bound / work lists,
// the substitution block's partner ids: ~75 KB -- asked from the runtime, not assumed), less a small reserve.  Static + dynamic beyond the limit
// fails in hipFuncSetAttribute or at launch with a raw HIP error; the caller turns it into a clear message.
size_t sel2_lds_budget(int device, bool wide) {
    static std::mutex mu;
    static size_t cached[2] = {0, 0};
    std::lock_guard<std::mutex> lk(mu);
    if (!cached[wide]) {
        hipFuncAttributes fa;
        const void *fn = wide ? reinterpret_cast<const void *>(&k_iter_select2<uint64_t>) : reinterpret_cast<const void *>(&k_iter_select2<uint32_t>);
        HIP_CHECK(hipFuncGetAttributes(&fa, fn));
        int limit = 0;
        HIP_CHECK(hipDeviceGetAttribute(&limit, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
        if (limit < 64 * 1024) limit = 64 * 1024;
        const size_t used = fa.sharedSizeBytes + 256;
        cached[wide] = (size_t)limit > used ? (size_t)limit - used : 1;
    }
    return cached[wide];
}
// words of the optional LDS area in which the substitution block combines the row bitmaps of a young chain: dropped (0: every wave ORs its words
// itself) when it does not fit beside the rest -- the kernel runs either way
int claim_words_for(const ChainJob &job, const Geometry &g, size_t budget) {
    const size_t words = ((size_t)g.rcap + 31) / 32, fixed = align_up(sel2_fixed_lds(job, g), 16);
    return words * 4 <= 64 * 1024 && fixed + words * 4 + 16 <= budget ? (int)words : 0;
}

// carve one chain's arrays; with base == nullptr only the size is computed
size_t carve_chain(unsigned char *base, const ChainJob &job, const Geometry &g, ChainDev &d) {
    Carver c(base);
    size_t cell = g.wide ? 8 : 4, entry = g.wide ? 16 : 4;
    size_t n_out = job.n_out;
    d.rlist = c.take<unsigned char>((size_t)g.rl_cap * entry);
    d.rowoff = c.take<da_u2>(g.rcap);
    d.rows = c.take<RowInfo>(g.rcap);
    d.stamp = c.take<uint32_t>(g.rcap);
    d.collist = c.take<unsigned long long>(n_out * (size_t)g.lcap);
    d.collen = c.take<int>(n_out);
    d.hkey = c.take<unsigned long long>(g.C);
    d.hrank = c.take<uint32_t>((size_t)g.C + ((size_t)g.C + 3) / 4);  // + the best-key indices, one byte per slot, right behind the ranks (hidx_ptr)
    d.hblk = c.take<unsigned char>((size_t)g.C << g.pb_log2);
    d.grec = c.take<GroupRec>(g.n_groups);
    d.mcol = c.take<int>(n_out);
    d.cmap = c.take<uint16_t>(n_out);
    d.colbits = c.take<uint32_t>(n_out * (size_t)((g.rcap + 31) / 32));
    d.pl_ids = c.take<uint32_t>(g.rcap);
    d.mA = c.take<unsigned char>(n_out * cell);
    d.mB = c.take<unsigned char>(n_out * cell);
    d.plist = c.take<unsigned long long>(g.rcap);
    d.sp_cnt = c.take<uint32_t>((size_t)6 * g.Kpad);
    d.picks = c.take<int4>(g.rcap);
    d.fin_row = c.take<uint32_t>(n_out * (size_t)g.lcap);
    d.fin_cell = c.take<unsigned long long>(n_out * (size_t)g.lcap);
    d.fin_count = c.take<uint32_t>(n_out);
    d.fin_start = c.take<uint32_t>(n_out + 1);
    d.pk_cap = g.pk_cap;
    d.pk_row = c.take<uint32_t>((size_t)g.pk_cap);
    d.pk_cell = c.take<unsigned long long>((size_t)g.pk_cap);
    d.pk_lat = c.take<float>(g.rcap);
    d.step_mant = c.take<uint32_t>((size_t)std::max(g.n_mant, 1));  // filled only when an input step is not a power of two (StepLog2)
    d.step_tab = c.take<float>((size_t)std::max(g.n_mant, 1) * 256);
    return align_up(c.off, 256);
}

}  // namespace

// Runs chains [all of one cell width] to completion.  `prep` holds the descriptors after k_prepare (inputs resident).
void HipBackend::run_chains(const ChainJob *jobs, ChainOut *outs, int n) {
    if (n <= 0) return;
    Impl &im = *impl_;
    HIP_CHECK(hipSetDevice(im.device));
    hipStream_t st = im.stream;
    auto t_begin = std::chrono::steady_clock::now();
    const bool verbose = std::getenv("DA4ML_HIP_VERBOSE") != nullptr;
    auto lap = [&, last = t_begin](const char *what) mutable {
        auto now = std::chrono::steady_clock::now();
        if (verbose) std::fprintf(stderr, "[da4ml_hip] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    };

    // ---- 1. inputs to the device, k_prepare
    std::vector<size_t> in_off(n);
    size_t in_bytes = 0;
    for (int i = 0; i < n; ++i) {
        in_off[i] = in_bytes;
        size_t e = (size_t)jobs[i].n_in * jobs[i].n_out;
        in_bytes += align_up(e * 4, 256) + align_up((size_t)jobs[i].n_in * 12, 256) + align_up((size_t)jobs[i].n_in * 4, 256) +
                    align_up(e * 4, 256) + align_up(jobs[i].n_in, 256) + align_up(jobs[i].n_out, 256);
    }
    unsigned char *io = static_cast<unsigned char *>(im.io_buf.get(std::max<size_t>(in_bytes, 256)));
    std::vector<ChainDev> desc(n);
    unsigned char *stage_ptr = static_cast<unsigned char *>(im.pinned_up.get(std::max<size_t>(in_bytes, 256)));  // pinned: the upload is one asynchronous DMA
    for (int i = 0; i < n; ++i) {
        const ChainJob &j = jobs[i];
        size_t e = (size_t)j.n_in * j.n_out, o = in_off[i];
        ChainDev &d = desc[i];
        std::memset(&d, 0, sizeof d);
        d.n_in = j.n_in;
        d.n_out = j.n_out;
        d.pn_out = j.n_out;
        d.method = j.method;
        d.adder_size = j.adder_size;
        d.carry_size = j.carry_size;
        d.kernel = reinterpret_cast<const float *>(io + o);
        o += align_up(e * 4, 256);
        d.qints = reinterpret_cast<const float *>(io + o);
        o += align_up((size_t)j.n_in * 12, 256);
        d.lats = reinterpret_cast<const float *>(io + o);
        o += align_up((size_t)j.n_in * 4, 256);
        d.xint = reinterpret_cast<int32_t *>(io + o);
        o += align_up(e * 4, 256);
        d.shift0 = reinterpret_cast<int8_t *>(io + o);
        o += align_up(j.n_in, 256);
        d.shift1 = reinterpret_cast<int8_t *>(io + o);
    }
    {  // the inputs into the pinned staging buffer, on a few host threads (16 MB for the 64 matrices of the benchmark)
        auto stage = [&](int i) {
            const ChainJob &j = jobs[i];
            const size_t e = (size_t)j.n_in * j.n_out;
            size_t o = in_off[i];
            std::memcpy(stage_ptr + o, j.kernel, e * 4);
            o += align_up(e * 4, 256);
            std::memcpy(stage_ptr + o, j.qints, (size_t)j.n_in * 12);
            o += align_up((size_t)j.n_in * 12, 256);
            std::memcpy(stage_ptr + o, j.lats, (size_t)j.n_in * 4);
        };
        const int workers = (int)std::min<size_t>({(size_t)n, (size_t)8, in_bytes / (1u << 20) + 1});
        std::atomic<int> next{0};
        auto work = [&] {
            for (int i = next++; i < n; i = next++) stage(i);
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < workers; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    HIP_CHECK(hipMemcpyAsync(io, stage_ptr, in_bytes, hipMemcpyHostToDevice, st));
    ChainDev *d_desc = static_cast<ChainDev *>(im.desc_buf.get(sizeof(ChainDev) * (size_t)n));
    HIP_CHECK(hipMemcpyAsync(d_desc, desc.data(), sizeof(ChainDev) * (size_t)n, hipMemcpyHostToDevice, st));
    int max_n_out = 0, max_n_in = 0;
    for (int i = 0; i < n; ++i) {
        max_n_out = std::max(max_n_out, jobs[i].n_out);
        max_n_in = std::max(max_n_in, jobs[i].n_in);
    }
    hipLaunchKernelGGL(k_prepare, dim3(n), dim3(256), (size_t)max_n_out * 4, st, d_desc);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(desc.data(), d_desc, sizeof(ChainDev) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));

    lap("upload + k_prepare");
    // ---- 2. geometry and arena
    // -log2f tables of non-power-of-two input steps (rare: the tracer's `variable * 3`), by the host libm, one row per distinct
    // mantissa -- as many as the inputs have (the reference takes log2 of any step, state_opr.cc:57); they stay alive until the
    // set-up stream has been synchronised below
    std::vector<StepLog2Host> step_tabs(n);
    for (int i = 0; i < n; ++i)
        if (jobs[i].adder_size >= 0 || jobs[i].carry_size >= 0) step_tabs[i].build(jobs[i].qints, jobs[i].n_in);  // (latency model off: steps are never looked at)
    std::vector<Geometry> geo(n);
    std::vector<size_t> a_off(n);
    size_t arena_bytes = 0;
    for (int i = 0; i < n; ++i) {
        const ChainDev &d = desc[i];
        Geometry &g = geo[i];
        g.n_bits = d.prep_nbits;
        if (g.n_bits > 30) throw std::runtime_error("kernel needs more than 30 CSD digits per entry (the reference overflows int32 there); unsupported");
        g.wide = g.n_bits > 12 || jobs[i].n_out > 256;  // the narrow list entry is col:8 | minus:12 | plus:12
        g.n_mant = (int)step_tabs[i].mant.size();
        g.K = key_count(g.n_bits);
        g.Kpad = (g.K + 3) & ~3;
        g.pb_log2 = 5;  // payload line of a pair block: 16-byte header + Kpad u16 counts, padded to a power of two
        while ((1 << g.pb_log2) < 16 + 2 * g.Kpad) ++g.pb_log2;
        long long D0 = d.prep_digits;
        // every greedy step removes at least one digit; typical chains need ~D0/8 steps
        long long steps = jobs[i].method == M_DUMMY || jobs[i].method < 0 ? 0 : std::max<long long>(16, (long long)(D0 * row_scale_ / 4));
        if (steps > D0) steps = std::max<long long>(D0, 1);
        g.rcap = jobs[i].n_in + (int)steps + 1;
        g.lcap = jobs[i].n_in + d.prep_maxdcol + 1;
        g.pk_cap = (int)std::min<long long>(D0 + 1, (long long)1 << 30);  // digits only ever disappear
        // row lists: the dense lists of the input rows + one entry per (new row, column), each holding at least one of
        // the digits that the substitutions move into new rows (at most D0 over a chain)
        const long long rl_want = (long long)jobs[i].n_in * jobs[i].n_out + D0 + jobs[i].n_out + 64;
        if (rl_want >= (1ll << REF_OFF_BITS) || g.rcap >= (1 << REF_ROW_BITS) || jobs[i].n_out >= (1 << REF_LEN_BITS))
            throw std::runtime_error("problem too large for the row-reference format (rows < 2^24, columns < 4096, list entries < 2^28)");
        g.rl_cap = (uint32_t)rl_want;
        // table capacity: blocks peak well above the initial pair count when rows are dense
        long long pairs0 = std::min<long long>((long long)jobs[i].n_in * (jobs[i].n_in + 1) / 2, std::max<long long>(d.prep_pairs, 1));
        double growth = std::max(4.0, jobs[i].n_in / 5.0);
        double want = std::max(1024.0, 0.8 * pairs0 * growth * im.table_scale);
        if (jobs[i].method == M_DUMMY) want = 64;
        g.C = pow2_ceil((uint64_t)want);
        g.gs_log2 = 8;
        while ((g.C >> g.gs_log2) > (uint32_t)MAX_GROUPS) ++g.gs_log2;
        if (g.gs_log2 > 14) throw std::runtime_error("pair table larger than 64M slots is not supported");
        if (g.C < 256) g.C = 256;
        g.n_groups = (int)(g.C >> g.gs_log2);
        ChainDev tmp;
        a_off[i] = arena_bytes;
        arena_bytes += carve_chain(nullptr, jobs[i], g, tmp);
    }
    // a batch whose arena does not fit into (most of) the free device memory is processed in two halves
    {
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        size_t budget = (size_t)(0.85 * (double)(free_b + im.arena.cap));
        if (const char *e = std::getenv("DA4ML_HIP_MEM_BUDGET_MB")) budget = (size_t)std::atoll(e) << 20;  // test hook
        if (arena_bytes > budget) {
            if (n == 1) throw std::runtime_error("a single chain needs " + std::to_string(arena_bytes >> 20) + " MiB of device memory, more than is free");
            const int half_n = n / 2;
            run_chains(jobs, outs, half_n);
            run_chains(jobs + half_n, outs + half_n, n - half_n);
            return;
        }
    }
    unsigned char *arena = static_cast<unsigned char *>(im.arena.get(arena_bytes));
    for (int i = 0; i < n; ++i) {
        ChainDev &d = desc[i];
        const Geometry &g = geo[i];
        carve_chain(arena + a_off[i], jobs[i], g, d);
        d.n_bits = g.n_bits;
        d.K = g.K;
        d.Kpad = g.Kpad;
        d.rcap = g.rcap;
        d.lcap = g.lcap;
        d.gs_log2 = g.gs_log2;
        d.n_groups = g.n_groups;
        d.C = g.C;
        d.cmask = g.C - 1;
        d.pb_log2 = g.pb_log2;
        d.rl_cap = g.rl_cap;
        d.rl_used = (uint32_t)jobs[i].n_in * (uint32_t)jobs[i].n_out;
        d.n_rows = jobs[i].n_in;
        d.claim_words = claim_words_for(jobs[i], g, sel2_lds_budget(im.device, g.wide));
        d.iter = 0;
        d.done = (jobs[i].method == M_DUMMY || jobs[i].method < 0) ? 1 : 0;
        d.cb_words = (g.rcap + 31) / 32;
    }
    for (int i = 0; i < n; ++i) {
        desc[i].n_step_mant = (int)step_tabs[i].mant.size();
        if (desc[i].n_step_mant) {
            HIP_CHECK(hipMemcpyAsync(const_cast<uint32_t *>(desc[i].step_mant), step_tabs[i].mant.data(), step_tabs[i].mant.size() * 4, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(const_cast<float *>(desc[i].step_tab), step_tabs[i].tab.data(), step_tabs[i].tab.size() * 4, hipMemcpyHostToDevice, st));
        }
    }
    HIP_CHECK(hipMemcpyAsync(d_desc, desc.data(), sizeof(ChainDev) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(im.d_done, 0, sizeof(unsigned int), st));
    // initial state of all chains in ONE launch (was six hipMemsetAsync per chain: 384 calls and 384 small kernels per batch)
    hipLaunchKernelGGL(k_init_state, dim3(128, n), dim3(256), 0, st, d_desc);
    HIP_CHECK(hipGetLastError());

    // ---- 3. launch per cell width (descriptors are grouped so that one launch covers a contiguous range)
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return geo[a].wide < geo[b].wide; });
    bool permuted = false;
    for (int i = 0; i < n; ++i) permuted |= order[i] != i;
    if (permuted) {
        std::vector<ChainDev> sorted(n);
        for (int i = 0; i < n; ++i) sorted[i] = desc[order[i]];
        HIP_CHECK(hipMemcpyAsync(d_desc, sorted.data(), sizeof(ChainDev) * (size_t)n, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    int n_narrow = 0;
    for (int i = 0; i < n; ++i) n_narrow += !geo[i].wide;
    struct Range {
        int first, count;
        bool wide;
    } ranges[2] = {{0, n_narrow, false}, {n_narrow, n - n_narrow, true}};
    int active = 0;
    for (int i = 0; i < n; ++i) active += desc[i].done ? 0 : 1;

    size_t sel_lds[2] = {0, 0}, upd_lds[2] = {0, 0}, pair_lds[2] = {0, 0};
    int upd_blocks[2] = {1, 1};
    long long max_pairs[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        int w = geo[i].wide;
        const size_t no = (size_t)jobs[i].n_out, cellb = geo[i].wide ? 8 : 4;
        const size_t s = align_up(sel2_fixed_lds(jobs[i], geo[i]) + (size_t)desc[i].claim_words * 4, 16);
        if (s > sel2_lds_budget(im.device, geo[i].wide))
            throw std::runtime_error("selection kernel needs " + std::to_string(s) + " bytes of dynamic LDS, the device leaves it " + std::to_string(sel2_lds_budget(im.device, geo[i].wide)) +
                                     " beside the kernel's static arrays (n_out too large)");
        sel_lds[w] = std::max(sel_lds[w], s);
        upd_lds[w] = std::max(upd_lds[w], align_up(2 * no * cellb + no * 6, 16) + align_up((size_t)UPD_WAVES * (QN * 3 + 1) * (size_t)geo[i].Kpad * 4, 16));  // UpdLds: hand-off tables | counters
        pair_lds[w] = std::max(pair_lds[w], (size_t)4 * geo[i].Kpad * 4);
        max_pairs[w] = std::max(max_pairs[w], (long long)jobs[i].n_in * (jobs[i].n_in + 1) / 2);
    }
    for (int w = 0; w < 2; ++w) {
        int cnt = ranges[w].count;
        if (cnt == 0) continue;
        // at least 2 and at most 64 blocks per chain
        upd_blocks[w] = std::max(2, std::min(64, (im.upd_total_blocks + cnt - 1) / std::max(cnt, 1)));
    }
    for (int w = 0; w < 2; ++w) {
        const Range &r = ranges[w];
        if (r.count == 0) continue;
        ChainDev *base = d_desc + r.first;
        dim3 colgrid((max_n_out + 3) / 4, r.count);
        dim3 pairgrid((unsigned)((max_pairs[w] + 3) / 4), r.count);
        if (!r.wide) {
            hipLaunchKernelGGL(k_init_cells<uint32_t>, colgrid, dim3(256), 0, st, base);
            hipLaunchKernelGGL(k_init_pairs<uint32_t>, pairgrid, dim3(256), pair_lds[w], st, base);
        } else {
            hipLaunchKernelGGL(k_init_cells<uint64_t>, colgrid, dim3(256), 0, st, base);
            hipLaunchKernelGGL(k_init_pairs<uint64_t>, pairgrid, dim3(256), pair_lds[w], st, base);
        }
        HIP_CHECK(hipGetLastError());
    }
    if (ranges[0].count)
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_select2<uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds[0]));
    if (ranges[1].count)
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_select2<uint64_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds[1]));

    HIP_CHECK(hipStreamSynchronize(st));
    lap("arena + init kernels");
    // ---- 4. greedy loop: two kernels per iteration.  The chains are split into up to four groups, each advancing
    // in lockstep on its own stream, so that the one-block-per-chain select kernel of one group overlaps the update
    // kernel of the others.
    EventGuard events;
    hipEvent_t ev0 = events.make(), ev1 = events.make();
    HIP_CHECK(hipEventRecord(ev0, st));
    HIP_CHECK(hipStreamSynchronize(st));  // set-up done before the group streams start
    struct Group {
        int first, count, w;  // descriptor range, cell width index
        hipStream_t stream;
    };
    std::vector<Group> groups;
    for (int w = 0; w < 2; ++w) {
        const Range &r = ranges[w];
        if (r.count == 0) continue;
        int parts = std::max(1, std::min(im.n_lanes, r.count / 8));
        if (ranges[0].count && ranges[1].count) parts = std::max(1, parts / 2);
        for (int p = 0; p < parts; ++p) {
            int lo = r.first + (int)((long long)r.count * p / parts), hi = r.first + (int)((long long)r.count * (p + 1) / parts);
            groups.push_back(Group{lo, hi - lo, w, im.lanes[groups.size() % Impl::MAX_LANES]});
        }
    }
    // One greedy iteration of one group = (select, update) on the group's stream.
    // `step` = the number of the lockstep iteration = the iteration count of every chain that has not finished (every launch pair advances
    // each of them by one): a kernel argument, because the search block of the selection must not read a field its sibling writes
    // DA4ML_HIP_STATS=1: k_iter_update tallies the blocks it finds / creates / deletes (da_timings' found / inserts, the "peak pair blocks" of da_result_stats);
    // read at every call, so that a benchmark can count on one pass and time the others
    const char *stats_env = std::getenv("DA4ML_HIP_STATS");
    const bool with_stats = stats_env && std::atoi(stats_env) != 0;
    auto launch_pair = [&](const Group &gr, hipEvent_t *se, int step) {
        ChainDev *base = d_desc + gr.first;
        const dim3 sel_grid((gr.count + 7) & ~7, 2);  // y = 0 search block, y = 1 substitution block
        if (se) HIP_CHECK(hipEventRecord(se[0], gr.stream));
        if (gr.w == 0)
            hipLaunchKernelGGL(k_iter_select2<uint32_t>, sel_grid, dim3(SEL2_THREADS), sel_lds[0], gr.stream, base, gr.count, im.d_done, step);
        else
            hipLaunchKernelGGL(k_iter_select2<uint64_t>, sel_grid, dim3(SEL2_THREADS), sel_lds[1], gr.stream, base, gr.count, im.d_done, step);
        if (se) HIP_CHECK(hipEventRecord(se[1], gr.stream));
        // (+ 2: the last two blocks of a chain write the six blocks of the pairs among the modified rows)
        const dim3 upd_grid((gr.count + 7) & ~7, upd_blocks[gr.w] + 2);
        if (gr.w == 0 && !with_stats)
            hipLaunchKernelGGL((k_iter_update<uint32_t, false>), upd_grid, dim3(UPD_THREADS), upd_lds[0], gr.stream, base, gr.count);
        else if (gr.w == 0)
            hipLaunchKernelGGL((k_iter_update<uint32_t, true>), upd_grid, dim3(UPD_THREADS), upd_lds[0], gr.stream, base, gr.count);
        else if (!with_stats)
            hipLaunchKernelGGL((k_iter_update<uint64_t, false>), upd_grid, dim3(UPD_THREADS), upd_lds[1], gr.stream, base, gr.count);
        else
            hipLaunchKernelGGL((k_iter_update<uint64_t, true>), upd_grid, dim3(UPD_THREADS), upd_lds[1], gr.stream, base, gr.count);
        if (se) HIP_CHECK(hipEventRecord(se[2], gr.stream));
    };
    // Windows of up to WINDOW_ITERS iterations x all groups are queued eagerly; one event-bracketed iteration per window
    // samples the kernel durations.  (A hipGraph replay of the window was re-measured in round 2: no gain.)
    constexpr int WINDOW_ITERS = 63, MAX_SAMPLES = 4096;
    std::vector<hipEvent_t> sample_ev;
    int n_samples = 0;
    long long launched_iters = 0, iter_cap = 0;
    for (int i = 0; i < n; ++i) iter_cap = std::max<long long>(iter_cap, geo[i].rcap - jobs[i].n_in + 2);
    const int poll_every = WINDOW_ITERS + 1;
    int pre_done = 0;
    for (int i = 0; i < n; ++i) pre_done += (jobs[i].method == M_DUMMY || jobs[i].method < 0) ? 1 : 0;
    // The done counter is read back ONE WINDOW BEHIND on a separate stream: after window w is queued, the poll stream
    // waits for every group's end-of-window event and copies the counter; the host looks at the copy of window w-1 only
    // after window w has been queued, so the queues never drain while the host decides whether to go on.
    hipEvent_t win_ev[2][Impl::MAX_LANES], copy_ev[2];
    for (int p = 0; p < 2; ++p) {
        copy_ev[p] = events.make(hipEventDisableTiming);
        for (size_t gi = 0; gi < groups.size(); ++gi) win_ev[p][gi] = events.make(hipEventDisableTiming);
    }
    im.h_done[0] = im.h_done[1] = 0;
    long long window = 0;
    double host_launch_ms = 0;  // host time spent queueing launches (not waiting for the device)
    struct DrainOnError {  // an exception between here and the end of the loop leaves launches queued: let them finish before the arena is reused
        const std::vector<Group> &g;
        hipStream_t poll;
        bool armed = true;
        ~DrainOnError() {
            if (!armed) return;
            for (const Group &gr : g) (void)hipStreamSynchronize(gr.stream);
            (void)hipStreamSynchronize(poll);
        }
    } drain{groups, im.poll_stream};
    // A second launching thread takes every other chain group: with the shorter kernels of round 5 one thread queueing all launches
    // (2.85 us each, 8 per lockstep iteration of 4 groups) is the bound for small problems (64x64: 22.9 of 24.6 us per iteration, measured).
    // It follows the windows of this thread: (first step, iterations) in, its groups' end-of-window events recorded out.
    struct Helper {
        std::atomic<long long> seq{0}, ack{0};
        std::atomic<bool> quit{false};
        long long first_step = 0;
        int iters = 0, parity = 0;
        std::exception_ptr err;
        std::thread th;
    } helper;
    const bool two_threads = im.launch_threads >= 2 && groups.size() >= 2;
    auto mine = [&](size_t gi, int who) { return !two_threads || (int)(gi & 1) == who; };
    auto window_launches = [&](int who, long long first_step, int iters, int parity, hipEvent_t *se0) {
        for (int it = 0; it < iters; ++it)
            for (size_t gi = 0; gi < groups.size(); ++gi)
                if (mine(gi, who)) launch_pair(groups[gi], it == 0 && gi == 0 ? se0 : nullptr, (int)(first_step + it));
        HIP_CHECK(hipGetLastError());
        for (size_t gi = 0; gi < groups.size(); ++gi)
            if (mine(gi, who)) HIP_CHECK(hipEventRecord(win_ev[parity][gi], groups[gi].stream));
    };
    if (two_threads)
        helper.th = std::thread([&] {
            long long seen = 0;
            try {
                HIP_CHECK(hipSetDevice(im.device));
                while (true) {
                    long long s;
                    unsigned polls = 0;
                    while ((s = helper.seq.load(std::memory_order_acquire)) == seen && !helper.quit.load(std::memory_order_acquire)) spin_wait_step(polls);
                    if (s == seen) break;
                    seen = s;
                    if (!helper.err) window_launches(1, helper.first_step, helper.iters, helper.parity, nullptr);
                    helper.ack.store(seen, std::memory_order_release);
                }
            } catch (...) {
                helper.err = std::current_exception();
                helper.ack.store(helper.seq.load(), std::memory_order_release);  // (whatever window was being served: the main thread rethrows)
                unsigned polls = 0;
                while (!helper.quit.load(std::memory_order_acquire)) {  // keep acknowledging until told to leave
                    helper.ack.store(helper.seq.load(), std::memory_order_release);
                    spin_wait_step(polls);
                }
            }
        });
    struct JoinHelper {
        Helper &h;
        ~JoinHelper() {
            h.quit.store(true, std::memory_order_release);
            if (h.th.joinable()) h.th.join();
        }
    } join_helper{helper};
    while (active > 0) {
        const auto t_q0 = std::chrono::steady_clock::now();
        if (launched_iters > iter_cap + 2 * poll_every) throw std::runtime_error("greedy loop did not terminate within its row capacity (internal error)");
        // small problems finish within a few iterations: start with short windows, double up to the full length
        const int this_window = (int)std::min<long long>(WINDOW_ITERS, (8ll << std::min<long long>(window, 8)) - 1);
        const int p = (int)(window & 1);
        if (two_threads) {
            helper.first_step = launched_iters, helper.iters = this_window + 1, helper.parity = p;
            helper.seq.fetch_add(1, std::memory_order_release);
        }
        // the first iteration of a window is the sampled one: the first group's kernels are bracketed by events on its stream
        hipEvent_t se[3];
        const bool sample = n_samples < MAX_SAMPLES;
        if (sample) {
            for (auto &e : se) {
                e = events.make();
                sample_ev.push_back(e);
            }
            ++n_samples;
        }
        window_launches(0, launched_iters, this_window + 1, p, sample ? se : nullptr);
        launched_iters += this_window + 1;
        if (two_threads) {
            const long long want = helper.seq.load(std::memory_order_relaxed);
            unsigned polls = 0;
            while (helper.ack.load(std::memory_order_acquire) != want) spin_wait_step(polls);
            if (helper.err) std::rethrow_exception(helper.err);
        }
        for (size_t gi = 0; gi < groups.size(); ++gi) HIP_CHECK(hipStreamWaitEvent(im.poll_stream, win_ev[p][gi], 0));
        HIP_CHECK(hipMemcpyAsync(&im.h_done[p], im.d_done, sizeof(unsigned int), hipMemcpyDeviceToHost, im.poll_stream));
        HIP_CHECK(hipEventRecord(copy_ev[p], im.poll_stream));
        host_launch_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_q0).count();
        if (window > 0) {
            HIP_CHECK(hipEventSynchronize(copy_ev[p ^ 1]));
            active = n - pre_done - (int)im.h_done[p ^ 1];
        }
        ++window;
    }
    for (const Group &gr : groups) HIP_CHECK(hipStreamSynchronize(gr.stream));
    HIP_CHECK(hipStreamSynchronize(im.poll_stream));
    drain.armed = false;
    HIP_CHECK(hipEventRecord(ev1, st));

    lap("greedy loop");
    // ---- 5. extraction and download
    for (int w = 0; w < 2; ++w) {
        const Range &r = ranges[w];
        if (r.count == 0) continue;
        dim3 colgrid((max_n_out + 3) / 4, r.count);
        if (!r.wide)
            hipLaunchKernelGGL(k_extract<uint32_t>, colgrid, dim3(256), 0, st, d_desc + r.first);
        else
            hipLaunchKernelGGL(k_extract<uint64_t>, colgrid, dim3(256), 0, st, d_desc + r.first);
    }
    hipLaunchKernelGGL(k_pack, dim3(n), dim3(256), 0, st, d_desc);
    HIP_CHECK(hipGetLastError());
    std::vector<ChainDev> fin(n);
    HIP_CHECK(hipMemcpyAsync(fin.data(), d_desc, sizeof(ChainDev) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    float loop_ms = 0;
    HIP_CHECK(hipEventElapsedTime(&loop_ms, ev0, ev1));
    for (int k = 0; k < n_samples; ++k) {
        float a = 0, b = 0;
        HIP_CHECK(hipEventElapsedTime(&a, sample_ev[3 * k], sample_ev[3 * k + 1]));
        HIP_CHECK(hipEventElapsedTime(&b, sample_ev[3 * k + 1], sample_ev[3 * k + 2]));
        im.timings.select_ms_sampled += a;
        im.timings.update_ms_sampled += b;
    }
    im.timings.samples += n_samples;
    im.timings.sampled_chain_launches += (double)n_samples * (groups.empty() ? 0 : groups[0].count);

    bool need_retry = false;
    for (int s = 0; s < n; ++s) {
        const ChainDev &d = fin[s];
        int i = order[s];
        ChainOut &o = outs[i];
        o = ChainOut{};
        o.error = d.error;
        o.unknown_method_hit = d.unknown_hit != 0;
        o.n_bits = d.n_bits;
        if (d.error == E_TABLE_CAPACITY || d.error == E_ROW_CAPACITY) need_retry = true;
    }
    if (need_retry && retry_depth_ < 4) {
        // some chain outgrew its arena: rerun the whole group with larger capacities (rare; sizes are heuristics)
        ++retry_depth_;
        im.timings.retries += 1;
        double keep_t = im.table_scale, keep_r = row_scale_;
        im.table_scale *= 4.0;
        row_scale_ *= 4.0;
        try {
            run_chains(jobs, outs, n);
        } catch (...) {
            im.table_scale = keep_t;
            row_scale_ = keep_r;
            --retry_depth_;
            throw;
        }
        im.table_scale = keep_t;
        row_scale_ = keep_r;
        --retry_depth_;
        return;
    }
    // Results: the seven arrays of every chain (column offsets, shifts, picks, row latencies, surviving digits) are gathered
    // into one contiguous device buffer by k_gather and leave in ONE copy into pinned memory (they used to leave one by one:
    // 448 copies per 64-chain batch in two synchronised phases).
    struct ResultOffsets {
        size_t st, s0, s1, pk, lat, row, cell;
        uint32_t total;
    };
    std::vector<ResultOffsets> roff(n);
    std::vector<GatherPiece> pieces;
    pieces.reserve((size_t)n * 7);
    size_t gather_bytes = 0;
    auto piece = [&](const void *src, size_t bytes) {
        const size_t at = gather_bytes;
        if (bytes) pieces.push_back(GatherPiece{src, (unsigned long long)at, (unsigned long long)bytes});
        gather_bytes += align_up(bytes, 64);
        return at;
    };
    for (int s = 0; s < n; ++s) {
        const ChainDev &d = fin[s];
        const ChainJob &j = jobs[order[s]];
        ResultOffsets &ro = roff[s];
        ro.total = d.error == E_OK ? d.pk_total : 0u;  // a failed chain delivers no digits (finalize_chain raises for it)
        ro.st = piece(d.fin_start, ((size_t)j.n_out + 1) * 4);
        ro.s0 = piece(d.shift0, (size_t)j.n_in);
        ro.s1 = piece(d.shift1, (size_t)j.n_out);
        ro.pk = piece(d.picks, (size_t)d.iter * sizeof(int4));
        ro.lat = piece(d.pk_lat, (size_t)d.n_rows * 4);
        ro.row = piece(d.pk_row, (size_t)ro.total * 4);
        ro.cell = piece(d.pk_cell, (size_t)ro.total * 8);
    }
    unsigned char *pin = static_cast<unsigned char *>(im.pinned.get(std::max<size_t>(gather_bytes, 64)));
    {
        unsigned char *gbuf = static_cast<unsigned char *>(im.gather_buf.get(std::max<size_t>(gather_bytes, 64)));
        GatherPiece *d_pieces = static_cast<GatherPiece *>(im.piece_buf.get(std::max<size_t>(pieces.size(), 1) * sizeof(GatherPiece)));
        HIP_CHECK(hipMemcpyAsync(d_pieces, pieces.data(), pieces.size() * sizeof(GatherPiece), hipMemcpyHostToDevice, st));
        for (size_t first = 0; first < pieces.size(); first += 32768) {  // grid.y is a 16-bit quantity
            const unsigned cnt = (unsigned)std::min<size_t>(32768, pieces.size() - first);
            hipLaunchKernelGGL(k_gather, dim3(16, cnt), dim3(256), 0, st, d_pieces + first, gbuf);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(pin, gbuf, gather_bytes, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    // out of the pinned buffer into the result vectors: the chains on a few host threads (45 MB per 64-chain batch)
    auto unpack = [&](int s) {
        const ChainDev &d = fin[s];
        const ChainJob &j = jobs[order[s]];
        const ResultOffsets &ro = roff[s];
        ChainOut &o = outs[order[s]];
        const uint32_t *cs = reinterpret_cast<const uint32_t *>(pin + ro.st);
        o.col_start.assign(cs, cs + j.n_out + 1);
        o.shift0.assign(reinterpret_cast<const int8_t *>(pin + ro.s0), reinterpret_cast<const int8_t *>(pin + ro.s0) + j.n_in);
        o.shift1.assign(reinterpret_cast<const int8_t *>(pin + ro.s1), reinterpret_cast<const int8_t *>(pin + ro.s1) + j.n_out);
        const int32_t *pk = reinterpret_cast<const int32_t *>(pin + ro.pk);
        o.picks.assign(pk, pk + (size_t)d.iter * 4);
        const float *lat = reinterpret_cast<const float *>(pin + ro.lat);
        o.row_lat.assign(lat, lat + d.n_rows);
        const uint32_t *pr = reinterpret_cast<const uint32_t *>(pin + ro.row);
        const unsigned long long *pc = reinterpret_cast<const unsigned long long *>(pin + ro.cell);
        o.dig_row.assign(pr, pr + ro.total);
        o.dig_cell.assign(pc, pc + ro.total);
    };
    {
        const int workers = (int)std::min<size_t>({(size_t)n, (size_t)8, gather_bytes / (1u << 20) + 1});
        std::atomic<int> next{0};
        std::exception_ptr err;
        std::mutex err_mu;
        auto work = [&] {
            try {
                for (int s = next++; s < n; s = next++) unpack(s);
            } catch (...) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (!err) err = std::current_exception();
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < workers; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (err) std::rethrow_exception(err);
    }
    for (int s = 0; s < n; ++s) {
        const ChainDev &d = fin[s];
        const int i = order[s];
        ChainOut &o = outs[i];
        o.stats.iterations = d.iter;
        o.stats.digits0 = d.prep_digits;
        o.stats.table_peak = d.live_peak;
        o.stats.scan_slots = (long long)d.st_rescans << d.gs_log2;
        o.stats.partners = (long long)d.st_partners;
        o.stats.matches = (long long)d.st_matches;
        for (int q = 0; q < 12; ++q) im.timings.phase_cycles[q] += (double)d.st_phase[q];
        for (int q = 0; q < 4; ++q) im.timings.search_cycles[q] += (double)d.st_qphase[q];
        im.timings.search_diag[0] += (double)d.st_qdiag[0], im.timings.search_diag[1] += (double)d.st_qdiag[1], im.timings.search_diag[2] += (double)d.st_qdiag[4];
        im.timings.search_diag[3] += (double)d.st_qdiag[5], im.timings.search_diag[4] = std::max(im.timings.search_diag[4], (double)d.st_qdiag[6]), im.timings.search_diag[5] += (double)d.st_qdiag[7], im.timings.search_diag[6] += (double)d.st_qdiag[8];
        im.timings.fast_steps += (long long)d.st_fast;
        im.timings.found += (long long)d.st_found;
        im.timings.inserts += (long long)d.st_inserts;
        im.timings.cell_reads += (long long)d.st_cells;
        im.timings.key_bytes += 2.0 * d.K * (double)(d.st_found + d.st_inserts);
        im.timings.cell_bytes += (geo[i].wide ? 8.0 : 4.0) * (double)d.st_cells;
        im.timings.iterations += d.iter;
        im.timings.rescans += (long long)d.st_rescans;
        // + the search block: bound, flags, tie word and lowered-value mark of every group per step (28 B), and per re-read group its ranks and ~2 slots' key and index
        im.timings.select_bytes += (double)d.st_sel_bytes + 32.0 * (double)d.n_groups * (double)d.iter + (double)d.st_rescans * ((double)(4u << d.gs_log2) + 24.0);
        im.timings.partners += (long long)d.st_partners;
        im.timings.table_bytes += (double)d.C * (8.0 + 4.0 + (double)(1 << d.pb_log2));
    }
    lap("extract + download + unpack");
    im.timings.loop_ms += loop_ms;
    im.timings.host_launch_ms += host_launch_ms;
    im.timings.lockstep_iters += launched_iters;
    im.timings.chains += n;
    im.timings.arena_bytes = std::max(im.timings.arena_bytes, (double)arena_bytes);
    im.timings.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    (void)max_n_in;
}

// ------------------------------------------------------------------------------------------------ column-sharded chain
namespace {

// da::ShardEngine on the GPU: one chain, the digits of the columns [c0, c1), a replica of the pair table.  Phases are
// kernel launches on the backend's stream, each followed by a stream synchronisation: the exchange between the phases
// (all-reduce of the buffers handed out here) is issued by the caller on its own stream / library.
class HipShardEngine : public ShardEngine {
  public:
    HipShardEngine(hipStream_t st, int device, const ChainJob &job, int c0, int c1, double table_scale, double row_scale)
        : st_(st), device_(device), job_(job), n_loc_(c1 - c0) {
        HIP_CHECK(hipSetDevice(device_));
        const size_t e = (size_t)job.n_in * job.n_out;
        // inputs + centred matrix of the WHOLE matrix (centring and digit width are global properties)
        const size_t kb = align_up(e * 4, 256), qb = align_up((size_t)job.n_in * 12, 256), lb = align_up((size_t)job.n_in * 4, 256);
        io_.get(2 * kb + qb + lb + align_up(job.n_in, 256) + align_up(job.n_out, 256) + 256);
        unsigned char *io = static_cast<unsigned char *>(io_.ptr);
        std::memset(&d_, 0, sizeof d_);
        d_.n_in = job.n_in;
        d_.n_out = n_loc_;
        d_.pn_out = job.n_out;
        d_.col0 = c0;
        d_.method = job.method;
        d_.adder_size = job.adder_size;
        d_.carry_size = job.carry_size;
        d_.kernel = reinterpret_cast<const float *>(io);
        d_.qints = reinterpret_cast<const float *>(io + kb);
        d_.lats = reinterpret_cast<const float *>(io + kb + qb);
        d_.xint = reinterpret_cast<int32_t *>(io + kb + qb + lb);
        d_.shift0 = reinterpret_cast<int8_t *>(io + 2 * kb + qb + lb);
        d_.shift1 = d_.shift0 + align_up(job.n_in, 256);
        HIP_CHECK(hipMemcpyAsync(io, job.kernel, e * 4, hipMemcpyHostToDevice, st_));
        HIP_CHECK(hipMemcpyAsync(io + kb, job.qints, (size_t)job.n_in * 12, hipMemcpyHostToDevice, st_));
        HIP_CHECK(hipMemcpyAsync(io + kb + qb, job.lats, (size_t)job.n_in * 4, hipMemcpyHostToDevice, st_));
        desc_.get(sizeof(ChainDev));
        dd_ = static_cast<ChainDev *>(desc_.ptr);
        push();
        hipLaunchKernelGGL(k_prepare, dim3(1), dim3(256), (size_t)job.n_out * 4, st_, dd_);
        HIP_CHECK(hipGetLastError());
        pull();
        // geometry: as HipBackend::run_chains, with the statistics of the whole matrix (the table is global)
        Geometry &g = geo_;
        g.n_bits = d_.prep_nbits;
        if (g.n_bits > 30) throw std::runtime_error("kernel needs more than 30 CSD digits per entry; unsupported");
        g.wide = g.n_bits > 12 || n_loc_ > 256;
        if (job.adder_size >= 0 || job.carry_size >= 0) step_tab_.build(job.qints, job.n_in);  // -log2f of non-power-of-two input steps (StepLog2), as in run_chains
        g.n_mant = (int)step_tab_.mant.size();
        g.K = key_count(g.n_bits);
        g.Kpad = (g.K + 3) & ~3;
        g.pb_log2 = 5;
        while ((1 << g.pb_log2) < 16 + 2 * g.Kpad) ++g.pb_log2;
        const long long D0 = d_.prep_digits;
        long long steps = std::max<long long>(16, (long long)(D0 * row_scale / 4));
        if (steps > D0) steps = std::max<long long>(D0, 1);
        g.rcap = job.n_in + (int)steps + 1;
        g.lcap = job.n_in + d_.prep_maxdcol + 1;
        g.pk_cap = (int)std::min<long long>(D0 + 1, (long long)1 << 30);
        const long long rl_want = (long long)job.n_in * n_loc_ + D0 + n_loc_ + 64;
        if (rl_want >= (1ll << REF_OFF_BITS) || g.rcap >= (1 << REF_ROW_BITS) || n_loc_ >= (1 << REF_LEN_BITS))
            throw std::runtime_error("problem too large for the row-reference format");
        g.rl_cap = (uint32_t)rl_want;
        const long long pairs0 = std::min<long long>((long long)job.n_in * (job.n_in + 1) / 2, std::max<long long>(d_.prep_pairs, 1));
        g.C = pow2_ceil((uint64_t)std::max(1024.0, 0.8 * pairs0 * std::max(4.0, job.n_in / 5.0) * table_scale));
        g.gs_log2 = 8;
        while ((g.C >> g.gs_log2) > (uint32_t)MAX_GROUPS) ++g.gs_log2;
        if (g.gs_log2 > 14) throw std::runtime_error("pair table larger than 64M slots is not supported");
        if (g.C < 256) g.C = 256;
        g.n_groups = (int)(g.C >> g.gs_log2);
        n_pairs_ = (long long)job.n_in * (job.n_in + 1) / 2;
        // arena: the chain's arrays (local column count) + the exchange buffers
        ChainJob local = job;
        local.n_out = n_loc_;
        ChainDev tmp;
        const size_t chain_bytes = carve_chain(nullptr, local, g, tmp);
        const size_t init_b = align_up((size_t)n_pairs_ * g.K * 4, 256), flag_b = align_up((((size_t)g.rcap + 3) / 4 + SHARD_TRAILER) * 4 + 64, 256),
                     uni_b = align_up((size_t)g.rcap * 4, 256), slab_b = align_up((size_t)(6 + 3 * (size_t)g.rcap) * g.K * 4, 256);
        arena_.get(chain_bytes + init_b + flag_b + uni_b + slab_b);
        unsigned char *a = static_cast<unsigned char *>(arena_.ptr);
        carve_chain(a, local, g, d_);
        d_.cs_init = reinterpret_cast<int32_t *>(a + chain_bytes);
        d_.cs_flags = reinterpret_cast<int32_t *>(a + chain_bytes + init_b);
        d_.cs_uni = reinterpret_cast<uint32_t *>(a + chain_bytes + init_b + flag_b);
        d_.cs_slab = reinterpret_cast<int32_t *>(a + chain_bytes + init_b + flag_b + uni_b);
        d_.n_bits = g.n_bits;
        d_.K = g.K;
        d_.Kpad = g.Kpad;
        d_.rcap = g.rcap;
        d_.lcap = g.lcap;
        d_.gs_log2 = g.gs_log2;
        d_.n_groups = g.n_groups;
        d_.C = g.C;
        d_.cmask = g.C - 1;
        d_.pb_log2 = g.pb_log2;
        d_.rl_cap = g.rl_cap;
        d_.rl_used = (uint32_t)job.n_in * (uint32_t)n_loc_;
        d_.n_rows = job.n_in;
        d_.claim_words = claim_words_for(local_job(), g, sel2_lds_budget(device_, g.wide));
        HIP_CHECK(hipMemsetAsync(d_.stamp, 0, sizeof(uint32_t) * (size_t)g.rcap, st_));
        HIP_CHECK(hipMemsetAsync(d_.hkey, 0xFF, sizeof(unsigned long long) * (size_t)g.C, st_));
        HIP_CHECK(hipMemsetAsync(d_.hrank, 0, sizeof(uint32_t) * (size_t)g.C, st_));
        HIP_CHECK(hipMemsetAsync(d_.grec, 0, sizeof(GroupRec) * (size_t)g.n_groups, st_));  // (bound 0 = nothing in the group: its flag is not looked at; the first entry that rises sets it)
        d_.cb_words = (g.rcap + 31) / 32;
        HIP_CHECK(hipMemsetAsync(d_.colbits, 0, sizeof(uint32_t) * (size_t)n_loc_ * d_.cb_words, st_));
        {
            d_.n_step_mant = (int)step_tab_.mant.size();
            if (d_.n_step_mant) {
                HIP_CHECK(hipMemcpyAsync(const_cast<uint32_t *>(d_.step_mant), step_tab_.mant.data(), step_tab_.mant.size() * 4, hipMemcpyHostToDevice, st_));
                HIP_CHECK(hipMemcpyAsync(const_cast<float *>(d_.step_tab), step_tab_.tab.data(), step_tab_.tab.size() * 4, hipMemcpyHostToDevice, st_));
            }
        }
        push();
        report_ = static_cast<volatile int *>(report_buf_.get(64));
        for (int q = 0; q < 5; ++q) report_[q] = 0;
        d_done_ = static_cast<unsigned int *>(done_buf_.get(sizeof(unsigned int)));  // (a member buffer: released also when a later check of this constructor throws)
        HIP_CHECK(hipMemsetAsync(d_done_, 0, sizeof(unsigned int), st_));
        dim3 colgrid((n_loc_ + 3) / 4, 1);
        if (!g.wide)
            hipLaunchKernelGGL(k_init_cells<uint32_t>, colgrid, dim3(256), 0, st_, dd_);
        else
            hipLaunchKernelGGL(k_init_cells<uint64_t>, colgrid, dim3(256), 0, st_, dd_);
        HIP_CHECK(hipGetLastError());
        sel_lds_ = align_up(sel2_fixed_lds(local_job(), g) + (size_t)d_.claim_words * 4, 16);
        if (sel_lds_ > sel2_lds_budget(device_, g.wide)) throw std::runtime_error("selection kernel needs more dynamic LDS than the device leaves beside its static arrays (n_out too large)");
        if (!g.wide)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_select2<uint32_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds_));
        else
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_iter_select2<uint64_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds_));
        part_lds_ = align_up(2 * (size_t)n_loc_ * (g.wide ? 8 : 4) + 4 * 3 * (size_t)g.Kpad * 4 + (size_t)n_loc_ * 2, 16);
        HIP_CHECK(hipStreamSynchronize(st_));
    }
    ~HipShardEngine() override { (void)hipSetDevice(device_); }
    bool on_device() const override { return true; }
    int n_keys() const override { return geo_.K; }
    int32_t *init_counts(int64_t &count) override {
        const dim3 grid((unsigned)((n_pairs_ + 3) / 4));
        if (!geo_.wide)
            hipLaunchKernelGGL(k_cs_init_counts<uint32_t>, grid, dim3(256), (size_t)4 * geo_.Kpad * 4, st_, dd_);
        else
            hipLaunchKernelGGL(k_cs_init_counts<uint64_t>, grid, dim3(256), (size_t)4 * geo_.Kpad * 4, st_, dd_);
        sync();
        count = n_pairs_ * geo_.K;
        return d_.cs_init;
    }
    void init_table() override {
        const dim3 grid((unsigned)((n_pairs_ + 3) / 4));
        if (!geo_.wide)
            hipLaunchKernelGGL(k_cs_init_table<uint32_t>, grid, dim3(256), 0, st_, dd_);
        else
            hipLaunchKernelGGL(k_cs_init_table<uint64_t>, grid, dim3(256), 0, st_, dd_);
        HIP_CHECK(hipGetLastError());
    }
    void set_stream_ordered(bool on) override { stream_ordered_ = on; }
    // The host reads NOTHING back between the kernels of a step except, once per step, the summed status trailer and the size
    // of the partner union (one synchronisation): the rows before a step are n_in + the steps taken, and a chain that stops
    // writes its zero flags and trailer on the device (select_body, shard_stop).  With a stream-ordered transport (the library's
    // RCCL transport: collectives queued on this stream) that is the only synchronisation of a step; a callback transport is
    // handed completed buffers, i.e. one synchronisation in front of each of its two calls.
    void select(int32_t *&flags, int64_t &fcount) override {
        fw_ = flag_words(job_.n_in + (int)steps_);  // rows before this step
        if (!stopped_) {
            // the two-block selection of the ordinary chains (search beside substitution, the pick known one step ahead): the table is a replica,
            // so every rank takes the same pick; the substitution block leaves flags and partial special-pair counts instead of a partner list
            if (!geo_.wide)
                hipLaunchKernelGGL((k_iter_select2<uint32_t, true>), dim3(1, 2), dim3(SEL2_THREADS), sel_lds_, st_, dd_, 1, d_done_, (int)steps_);
            else
                hipLaunchKernelGGL((k_iter_select2<uint64_t, true>), dim3(1, 2), dim3(SEL2_THREADS), sel_lds_, st_, dd_, 1, d_done_, (int)steps_);
            HIP_CHECK(hipGetLastError());
        } else {  // (not reached by ShardedBackend, which leaves the loop with the step that stopped; kept well-defined)
            const int32_t tr[SHARD_TRAILER] = {1, 0, 0};
            HIP_CHECK(hipMemsetAsync(d_.cs_flags, 0, (size_t)fw_ * 4, st_));
            HIP_CHECK(hipMemcpyAsync(d_.cs_flags + fw_, tr, sizeof tr, hipMemcpyHostToDevice, st_));
            sync();
        }
        if (!stream_ordered_) sync();
        flags = d_.cs_flags;
        fcount = fw_ + SHARD_TRAILER;
    }
    int32_t *partial(int64_t &scount, int32_t status[SHARD_TRAILER]) override {
        // the summed status and the size of the union come from the device itself: k_cs_union writes them into pinned host memory, the
        // host waits for the step's sequence number -- no copies, no stream synchronisation (the launch is checked; a device fault
        // surfaces through the bounded wait's fall-back synchronisation)
        const int seq = ++report_seq_;
        hipLaunchKernelGGL(k_cs_union, dim3(1), dim3(1024), 0, st_, dd_, report_, seq, (int)fw_);  // (on summed flags that are all zero when every rank has stopped: an empty union)
        HIP_CHECK(hipGetLastError());
        {
            unsigned polls = 0, tries = 0;
            while (__atomic_load_n(&report_[0], __ATOMIC_ACQUIRE) != seq) {
                if (++tries > (1u << 16)) {  // (seconds of polling, most of it asleep) let the runtime wait -- and report a fault, if that is what it is
                    sync();
                    if (__atomic_load_n(&report_[0], __ATOMIC_ACQUIRE) != seq) throw std::runtime_error("column-sharded chain: the device did not report the step's status");
                    break;
                }
                spin_wait_step(polls);
            }
        }
        const int nuni = report_[1];
        for (int q = 0; q < SHARD_TRAILER; ++q) trailer_[q] = report_[2 + q];
        for (int q = 0; q < SHARD_TRAILER; ++q) status[q] = trailer_[q];
        scount = 0;
        if (status[0] != 0) {
            stopped_ = true;
            return nullptr;
        }
        nuni_ = nuni;
        if (nuni > 0) {
            const dim3 grid((unsigned)((nuni + 3) / 4));
            if (!geo_.wide)
                hipLaunchKernelGGL(k_cs_partial<uint32_t>, grid, dim3(256), part_lds_, st_, dd_);
            else
                hipLaunchKernelGGL(k_cs_partial<uint64_t>, grid, dim3(256), part_lds_, st_, dd_);
        }
        if (!stream_ordered_) sync();
        scount = (int64_t)(6 + 3 * (int64_t)nuni) * geo_.K;
        return d_.cs_slab;
    }
    void apply() override {
        const dim3 grid((unsigned)((nuni_ + 6 + 3) / 4));
        if (!geo_.wide)
            hipLaunchKernelGGL(k_cs_apply<uint32_t>, grid, dim3(256), 0, st_, dd_);
        else
            hipLaunchKernelGGL(k_cs_apply<uint64_t>, grid, dim3(256), 0, st_, dd_);
        HIP_CHECK(hipGetLastError());
        ++steps_;
    }
    void finish(ChainOut &o) override {
        dim3 colgrid((n_loc_ + 3) / 4, 1);
        if (!geo_.wide)
            hipLaunchKernelGGL(k_extract<uint32_t>, colgrid, dim3(256), 0, st_, dd_);
        else
            hipLaunchKernelGGL(k_extract<uint64_t>, colgrid, dim3(256), 0, st_, dd_);
        hipLaunchKernelGGL(k_pack, dim3(1), dim3(256), 0, st_, dd_);
        pull();
        o = ChainOut{};
        o.error = d_.error;
        o.unknown_method_hit = d_.unknown_hit != 0;
        o.n_bits = d_.n_bits;
        const size_t iters = (size_t)d_.iter;
        o.shift0.resize(job_.n_in);
        o.shift1.resize(job_.n_out);
        o.picks.resize(iters * 4);
        o.row_lat.resize((size_t)d_.n_rows);
        o.col_start.resize((size_t)n_loc_ + 1);
        HIP_CHECK(hipMemcpyAsync(o.shift0.data(), d_.shift0, job_.n_in, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipMemcpyAsync(o.shift1.data(), d_.shift1, job_.n_out, hipMemcpyDeviceToHost, st_));
        if (iters) HIP_CHECK(hipMemcpyAsync(o.picks.data(), d_.picks, iters * sizeof(int4), hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipMemcpyAsync(o.row_lat.data(), d_.pk_lat, (size_t)d_.n_rows * 4, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipMemcpyAsync(o.col_start.data(), d_.fin_start, ((size_t)n_loc_ + 1) * 4, hipMemcpyDeviceToHost, st_));
        sync();
        const size_t total = o.col_start[n_loc_];
        o.dig_row.resize(total);
        std::vector<unsigned long long> cells(total);
        if (total) {
            HIP_CHECK(hipMemcpyAsync(o.dig_row.data(), d_.pk_row, total * 4, hipMemcpyDeviceToHost, st_));
            HIP_CHECK(hipMemcpyAsync(cells.data(), d_.pk_cell, total * 8, hipMemcpyDeviceToHost, st_));
            sync();
        }
        o.dig_cell.assign(cells.begin(), cells.end());
        o.stats.iterations = d_.iter;
        o.stats.digits0 = d_.prep_digits;
        o.stats.table_peak = d_.live_peak;
        o.stats.partners = (long long)d_.st_partners;
        o.stats.matches = (long long)d_.st_matches;
        o.stats.scan_slots = (long long)d_.st_rescans << d_.gs_log2;
    }

  private:
    ChainJob local_job() const {  // the chain as this rank holds it: n_loc_ columns
        ChainJob j = job_;
        j.n_out = n_loc_;
        return j;
    }
    void sync() {
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st_));
    }
    void push() { HIP_CHECK(hipMemcpyAsync(dd_, &d_, sizeof d_, hipMemcpyHostToDevice, st_)); }
    void pull() {
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(&d_, dd_, sizeof d_, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
    }
    hipStream_t st_;
    StepLog2Host step_tab_;
    int64_t fw_ = 0;                      // flag words of the current step (the status trailer follows them)
    long long steps_ = 0;                 // greedy steps applied so far (rows = n_in + steps_)
    bool stopped_ = false, stream_ordered_ = false;
    int32_t trailer_[SHARD_TRAILER] = {0, 0, 0};
    int device_;
    ChainJob job_;
    int n_loc_;  // columns of this rank (the first of them is d_.col0)
    ChainDev d_;
    ChainDev *dd_ = nullptr;
    Geometry geo_;
    DeviceBuffer io_, desc_, arena_;
    DeviceBuffer done_buf_;
    unsigned int *d_done_ = nullptr;
    PinnedBuffer report_buf_;          // {sequence number, union size, status[3]} written by k_cs_union (mapped pinned memory)
    volatile int *report_ = nullptr;
    int report_seq_ = 0;
    long long n_pairs_ = 0;
    int nuni_ = 0;
    size_t sel_lds_ = 0, part_lds_ = 0;
};

}  // namespace

std::unique_ptr<ShardEngine> HipBackend::make_shard_engine(const ChainJob &job, int c0, int c1, double capacity_scale) {
    return std::unique_ptr<ShardEngine>(new HipShardEngine(impl_->stream, impl_->device, job, c0, c1, impl_->table_scale * capacity_scale, row_scale_ * capacity_scale));
}

void HipBackend::column_distances(const int32_t *aug, int n_in, int W, int64_t *d0, int64_t *d1) {
    Impl &im = *impl_;
    HIP_CHECK(hipSetDevice(im.device));
    hipStream_t st = im.stream;
    size_t a_bytes = align_up((size_t)n_in * W * 4, 256), d_bytes = align_up((size_t)W * W * 8, 256);
    unsigned char *buf = static_cast<unsigned char *>(im.io_buf.get(a_bytes + 2 * d_bytes));
    HIP_CHECK(hipMemcpyAsync(buf, aug, (size_t)n_in * W * 4, hipMemcpyHostToDevice, st));
    auto *dd0 = reinterpret_cast<long long *>(buf + a_bytes), *dd1 = reinterpret_cast<long long *>(buf + a_bytes + d_bytes);
    EventGuard events;
    hipEvent_t e0 = events.make(), e1 = events.make();
    HIP_CHECK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_col_dist, dim3((W + 15) / 16, (W + 15) / 16), dim3(16, 16), 0, st, reinterpret_cast<const int32_t *>(buf), n_in, W, dd0, dd1);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(e1, st));
    HIP_CHECK(hipMemcpyAsync(d0, dd0, (size_t)W * W * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(d1, dd1, (size_t)W * W * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    im.timings.dist_ms += ms;
    im.timings.dist_calls += 1;
}

int HipBackend::int_to_csd(const int32_t *x, int64_t n, std::vector<int8_t> &csd) {
    Impl &im = *impl_;
    HIP_CHECK(hipSetDevice(im.device));
    hipStream_t st = im.stream;
    size_t xb = align_up(std::max<size_t>((size_t)n * 4, 4), 256);
    // two-step: global |max| -> N, then the digits
    unsigned char *buf = static_cast<unsigned char *>(im.io_buf.get(xb + 256 + (size_t)n * 33));
    auto *dx = reinterpret_cast<int32_t *>(buf);
    auto *dmax = reinterpret_cast<unsigned int *>(buf + xb);
    auto *dout = reinterpret_cast<int8_t *>(buf + xb + 256);
    HIP_CHECK(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(dmax, 0, 4, st));
    unsigned int mx = 0;
    if (n > 0) {
        int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
        hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, st, dx, (long long)n, dmax);
        HIP_CHECK(hipMemcpyAsync(&mx, dmax, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    int N = csd_width(mx);
    csd.assign((size_t)n * N, 0);
    if (n > 0) {
        hipLaunchKernelGGL(k_naf_digits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dx, (long long)n, N, dout);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(csd.data(), dout, (size_t)n * N, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    return N;
}

int HipBackend::csd_decompose(const float *kernel, int n_in, int n_out, bool center, std::vector<int8_t> &csd,
                              std::vector<int8_t> &s0, std::vector<int8_t> &s1) {
    // centring on the device through k_prepare of a one-chain batch, then the digit kernel
    Impl &im = *impl_;
    HIP_CHECK(hipSetDevice(im.device));
    hipStream_t st = im.stream;
    size_t e = (size_t)n_in * n_out;
    std::vector<int32_t> xi(e);
    if (center) {
        size_t kb = align_up(e * 4, 256), qb = align_up((size_t)n_in * 12, 256);
        unsigned char *buf = static_cast<unsigned char *>(im.io_buf.get(2 * kb + qb + align_up(n_in, 256) + align_up(n_out, 256) + 512));
        ChainDev d;
        std::memset(&d, 0, sizeof d);
        d.n_in = n_in;
        d.n_out = n_out;
        d.pn_out = n_out;
        d.kernel = reinterpret_cast<const float *>(buf);
        d.qints = reinterpret_cast<const float *>(buf + kb);
        d.xint = reinterpret_cast<int32_t *>(buf + kb + qb);
        d.shift0 = reinterpret_cast<int8_t *>(buf + 2 * kb + qb);
        d.shift1 = d.shift0 + align_up(n_in, 256);
        std::vector<float> ones((size_t)n_in * 3, 1.0f);  // no row is treated as dead here
        HIP_CHECK(hipMemcpyAsync(buf, kernel, e * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(buf + kb, ones.data(), ones.size() * 4, hipMemcpyHostToDevice, st));
        ChainDev *dd = static_cast<ChainDev *>(im.desc_buf.get(sizeof(ChainDev)));
        HIP_CHECK(hipMemcpyAsync(dd, &d, sizeof d, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_prepare, dim3(1), dim3(256), (size_t)n_out * 4, st, dd);
        HIP_CHECK(hipGetLastError());
        s0.resize(n_in);
        s1.resize(n_out);
        HIP_CHECK(hipMemcpyAsync(xi.data(), d.xint, e * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(s0.data(), d.shift0, n_in, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(s1.data(), d.shift1, n_out, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    } else {
        s0.assign(n_in, 0);
        s1.assign(n_out, 0);
        for (size_t k = 0; k < e; ++k) xi[k] = (int32_t)kernel[k];
    }
    return int_to_csd(xi.data(), (int64_t)e, csd);
}

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // namespace gpu
}  // namespace da
