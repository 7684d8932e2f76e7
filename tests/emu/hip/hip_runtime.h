// TEST INFRASTRUCTURE -- not part of the product.  A host-side stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel
// sources of da4ml_amd/csrc be compiled as plain C++ and executed on the CPU (tests/emu/Makefile puts this directory first
// on the include path): every GPU thread is a fiber, a block runs its wavefronts one after the other, wave-level
// operations (ballot, shuffles, DPP, readlane, wave barriers) are rendezvous points of the lanes that reach the same
// call site, __syncthreads is a rendezvous of the block.  Blocks and kernels run sequentially, so the emulation checks
// indexing and logic of the kernels and of the host code around them -- not their races and not their speed.
// The HIP runtime calls the product uses are mapped to malloc / memcpy / immediate execution.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <x86intrin.h>

// ------------------------------------------------------------------------------------------------ language surface
#define __global__
#define __device__
#define __host__
#define __constant__
// Blocks run one at a time on one thread: function-local statics are the block's LDS variables.  They live in a section of
// their own so that a race-detecting build can declare them new memory at every block start (LDS is per block).
#define __shared__ static __attribute__((section("hipemu_lds")))
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) x
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

struct int4 {
    int x, y, z, w;
};
#define amdgpu_num_sgpr(n) unused  // register-budget attribute of the update kernel: meaningless on the host
#define amdgpu_waves_per_eu(a, b) unused  // likewise (selection kernel)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu {
constexpr int WAVE = 64;
struct Snap {  // what a wave-level rendezvous delivers to every participant
    uint64_t mask;        // participating lanes
    uint64_t val[WAVE];   // their operands
};
struct ThreadCtx {
    dim3 tid, bid, bdim, gdim;
};
extern thread_local ThreadCtx *g_ctx;
void wave_exchange(uint64_t my_val, const void *site, Snap &out);  // blocks the calling lane until the group is complete
void block_barrier();
void sleep_yield();
unsigned char *dyn_shared();
void launch_impl(dim3 grid, dim3 block, size_t lds, void (*tramp)(void *), void *closure);
template <class F> void launch(dim3 grid, dim3 block, size_t lds, F &&f) {
    using Fn = typename std::remove_reference<F>::type;
    launch_impl(grid, block, lds, [](void *p) { (*static_cast<Fn *>(p))(); }, &f);
}
inline int lane() { return (int)((g_ctx->tid.x + g_ctx->bdim.x * (g_ctx->tid.y + g_ctx->bdim.y * g_ctx->tid.z)) & (WAVE - 1)); }
template <class T> inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8, "wave operand wider than 64 bits");
    uint64_t b = 0;
    std::memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T> inline T from_bits(uint64_t b) {
    T v;
    std::memcpy(&v, &b, sizeof(T));
    return v;
}
#define HIPEMU_SITE __builtin_return_address(0)
}  // namespace hipemu

#define threadIdx (::hipemu::g_ctx->tid)
#define blockIdx (::hipemu::g_ctx->bid)
#define blockDim (::hipemu::g_ctx->bdim)
#define gridDim (::hipemu::g_ctx->gdim)
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    ::hipemu::launch(dim3(grid), dim3(block), (size_t)(lds), [&]() { kernel(__VA_ARGS__); })

// ------------------------------------------------------------------------------------------------ wave-level operations
// Each of these is ONE non-inlined function, so that its return address identifies the (inlined) call site in the kernel.
__attribute__((noinline)) inline unsigned long long __ballot(int pred) {
    hipemu::Snap s;
    hipemu::wave_exchange(pred ? 1 : 0, HIPEMU_SITE, s);
    unsigned long long m = 0;
    for (int l = 0; l < hipemu::WAVE; ++l)
        if ((s.mask >> l & 1) && s.val[l]) m |= 1ull << l;
    return m;
}
namespace hipemu {
__attribute__((noinline)) inline uint64_t shfl_bits(uint64_t v, int src, const void *site) {
    Snap s;
    wave_exchange(v, site, s);
    src &= WAVE - 1;
    return (s.mask >> src & 1) ? s.val[src] : 0;  // ds_bpermute: a disabled source lane yields 0
}
}  // namespace hipemu
template <class T> __attribute__((noinline)) inline T __shfl(T v, int src) { return hipemu::from_bits<T>(hipemu::shfl_bits(hipemu::to_bits(v), src, HIPEMU_SITE)); }
template <class T> __attribute__((noinline)) inline T __shfl_up(T v, unsigned delta) {
    const int l = hipemu::lane();
    const uint64_t r = hipemu::shfl_bits(hipemu::to_bits(v), l - (int)delta < 0 ? l : l - (int)delta, HIPEMU_SITE);
    return l - (int)delta < 0 ? v : hipemu::from_bits<T>(r);
}
template <class T> __attribute__((noinline)) inline T __shfl_down(T v, unsigned delta) {
    const int l = hipemu::lane();
    const uint64_t r = hipemu::shfl_bits(hipemu::to_bits(v), l + (int)delta >= hipemu::WAVE ? l : l + (int)delta, HIPEMU_SITE);
    return l + (int)delta >= hipemu::WAVE ? v : hipemu::from_bits<T>(r);
}
template <class T> __attribute__((noinline)) inline T __shfl_xor(T v, int mask) { return hipemu::from_bits<T>(hipemu::shfl_bits(hipemu::to_bits(v), hipemu::lane() ^ mask, HIPEMU_SITE)); }

__attribute__((noinline)) inline int __any(int pred) {
    hipemu::Snap s;
    hipemu::wave_exchange(pred ? 1 : 0, HIPEMU_SITE, s);
    for (int l = 0; l < hipemu::WAVE; ++l)
        if ((s.mask >> l & 1) && s.val[l]) return 1;
    return 0;
}
__attribute__((noinline)) inline int __all(int pred) {
    hipemu::Snap s;
    hipemu::wave_exchange(pred ? 1 : 0, HIPEMU_SITE, s);
    for (int l = 0; l < hipemu::WAVE; ++l)
        if ((s.mask >> l & 1) && !s.val[l]) return 0;
    return 1;
}
__attribute__((noinline)) inline int __builtin_amdgcn_readlane(int v, int src_lane) {  // v_readlane ignores EXEC; a lane outside the group reads as 0
    hipemu::Snap s;
    hipemu::wave_exchange((uint32_t)v, HIPEMU_SITE, s);
    src_lane &= hipemu::WAVE - 1;
    return (s.mask >> src_lane & 1) ? (int)(uint32_t)s.val[src_lane] : 0;
}
__attribute__((noinline)) inline int __builtin_amdgcn_readfirstlane(int v) {
    hipemu::Snap s;
    hipemu::wave_exchange((uint32_t)v, HIPEMU_SITE, s);
    return (int)(uint32_t)s.val[__builtin_ctzll(s.mask)];
}
// v_mov_b32_dpp: the control words the kernels use (row_shr:n, row_bcast:15, row_bcast:31), row and bank masks, bound_ctrl
__attribute__((noinline)) inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    hipemu::Snap s;
    hipemu::wave_exchange((uint32_t)src, HIPEMU_SITE, s);
    const int l = hipemu::lane(), row = l >> 4, in_row = l & 15;
    if (!(row_mask >> row & 1) || !(bank_mask >> (in_row >> 2) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11F) {  // row_shr:n
        const int n = ctrl & 0xF;
        from = in_row >= n ? l - n : -1;
    } else if (ctrl >= 0x101 && ctrl <= 0x10F) {  // row_shl:n
        const int n = ctrl & 0xF;
        from = in_row + n < 16 ? l + n : -1;
    } else if (ctrl >= 0 && ctrl <= 0xFF) {  // quad_perm:[a,b,c,d] -- within every group of four lanes
        from = (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3);
    } else if (ctrl == 0x140) {  // row_mirror
        from = (l & ~15) + (15 - in_row);
    } else if (ctrl == 0x141) {  // row_half_mirror -- within every group of eight lanes
        from = (l & ~7) + (7 - (l & 7));
    } else if (ctrl == 0x142) {  // row_bcast:15 -- lane 15 of the previous row
        from = row >= 1 ? row * 16 - 1 : -1;
    } else if (ctrl == 0x143) {  // row_bcast:31 -- lane 31 for rows 2 and 3
        from = row >= 2 ? 31 : -1;
    } else {
        std::abort();  // a DPP control word this stand-in does not model
    }
    if (from < 0 || !(s.mask >> from & 1)) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)s.val[from];
}
__attribute__((noinline)) inline void __builtin_amdgcn_wave_barrier() {  // the lanes of a real wave are in lockstep here
    hipemu::Snap s;
    hipemu::wave_exchange(0, HIPEMU_SITE, s);
}
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_s_sleep(int) { hipemu::sleep_yield(); }  // a spin-wait: let the other wavefronts of the block run
inline void __builtin_amdgcn_sched_barrier(int) {}  // instruction-scheduling fence: no meaning on the host
inline void __builtin_amdgcn_s_setprio(int) {}
inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }  // HW_REG_XCC_ID: one emulated XCD
#define __builtin_amdgcn_fence(order, scope) ((void)0)         // blocks run one after the other: every store is visible
inline void __syncthreads() { hipemu::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_system() {}
inline void __threadfence_block() {}

// ------------------------------------------------------------------------------------------------ scalar device functions
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __mul24(int a, int b) { return a * b; }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline float __fmul_rn(float a, float b) { return a * b; }  // the emulation library is built with -ffp-contract=off
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __int_as_float(int v) { return hipemu::from_bits<float>((uint32_t)v); }
inline int __float_as_int(float v) { return (int)(uint32_t)hipemu::to_bits(v); }
inline long long clock64() { return (long long)__rdtsc(); }
inline long long wall_clock64() { return (long long)__rdtsc(); }
template <class A, class B> inline typename std::common_type<A, B>::type max(A a, B b) { return a > b ? a : b; }
template <class A, class B> inline typename std::common_type<A, B>::type min(A a, B b) { return a < b ? a : b; }

// atomics: kernels run one thread at a time; real atomic builtins all the same, so that a ThreadSanitizer build of the
// library (race detection, emu_runtime.cc) knows which accesses are atomic
template <class T, class U> inline T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> inline T atomicSub(T *p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> inline T atomicAnd(T *p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> inline T atomicExch(T *p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> inline T atomicMax(T *p, U v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v > o && !__atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return o;
}
template <class T, class U> inline T atomicMin(T *p, U v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v < o && !__atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return o;
}
template <class T, class U, class V> inline T atomicCAS(T *p, U cmp, V v) {
    T o = (T)cmp;
    __atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return o;
}
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_RELAXED)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define __hip_atomic_fetch_max(p, v, order, scope) atomicMax((p), (v))

// ------------------------------------------------------------------------------------------------ runtime API
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
struct hipemu_stream;
struct hipemu_event;
typedef hipemu_stream *hipStream_t;
typedef hipemu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
constexpr unsigned hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2;

inline const char *hipGetErrorString(hipError_t) { return "emulated HIP runtime"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) {  // HIPEMU_DEVICES: the multi-rank tests give every rank "its own" emulated device
    const char *e = std::getenv("HIPEMU_DEVICES");
    *n = e && std::atoi(e) > 0 ? std::atoi(e) : 1;
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
struct hipDeviceProp_t {
    int multiProcessorCount;
};
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 3; return hipSuccess; }  // three workgroups in the persistent grid
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)3 << 30; *total_b = (size_t)4 << 30; return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { return posix_memalign(p, 256, n ? n : 1) == 0 ? hipSuccess : 2; }  // exact size: a sanitizer build sees overruns
template <class T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
template <class T> inline hipError_t hipMemcpyToSymbol(T &sym, const void *s, size_t n) { std::memcpy(&sym, s, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
struct hipFuncAttributes { size_t sharedSizeBytes = 0; int numRegs = 0; };
// HIPEMU_STATIC_LDS (test hook): the static __shared__ bytes the emulated kernels report -- lets the CPU suite drive the product's LDS budget check
inline hipError_t hipFuncGetAttributes(hipFuncAttributes *a, const void *) { *a = hipFuncAttributes{}; if (const char *e = getenv("HIPEMU_STATIC_LDS")) a->sharedSizeBytes = (size_t)atoll(e); return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMaxSharedMemoryPerBlock = 8 };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 160 * 1024; return hipSuccess; }
