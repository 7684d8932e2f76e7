// What does the boundary between the greedy loop's two kernels cost, and does the SHAPE of the second kernel's workgroups matter?  (round 6: a chain sees
// ~2 us between its selection and its update, but ~4.4 us between its update and the next selection -- 1024-thread workgroups with ~50 KB of LDS.)
// Pairs (U, S) on one stream, timed host side over many pairs; every kernel's blocks spin for `work` ticks of the 100 MHz clock so that the pair has a known floor:
//   U = 672 workgroups x 256 threads (k_iter_update's grid at batch 64 / 4 groups); S = the selection in several shapes.
//   hipcc --offload-arch=gfx950 -O3 -o gap_probe gap_probe.hip && ./gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)
__global__ void k_u(unsigned *out, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (ticks < 0) out[0] = 1;
}
template <int LDS_STATIC> __global__ void k_s(unsigned *out, long long ticks) {
    __shared__ unsigned s_pad[LDS_STATIC / 4 + 1];
    extern __shared__ unsigned char dyn[];
    if (threadIdx.x == 0) { s_pad[0] = 1; dyn[0] = 1; }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (ticks < 0) out[0] = s_pad[0] + dyn[0];
}
template <class KS> double pair_us(KS ks, int s_blocks, int s_threads, size_t s_dyn, int u_blocks, unsigned *d, hipStream_t st, long long ticks, int n) {
    for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_u, dim3(u_blocks), dim3(256), 0, st, d, ticks); hipLaunchKernelGGL(ks, dim3(s_blocks), dim3(s_threads), s_dyn, st, d, ticks); }
    CK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_u, dim3(u_blocks), dim3(256), 0, st, d, ticks); hipLaunchKernelGGL(ks, dim3(s_blocks), dim3(s_threads), s_dyn, st, d, ticks); }
    CK(hipStreamSynchronize(st));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
// (round 6, second question) what does the END of a kernel cost as a function of what it wrote?  U2: every thread issues `per_thread` writes to pseudo-random
// 128-byte lines of a 1 GB buffer -- plain 4-byte stores, fire-and-forget atomicMax, or 16-byte stores -- then spins like U; S follows.
template <int MODE> __global__ void k_u2(unsigned *buf, unsigned mask_lines, int per_thread, long long ticks) {
    const long long t0 = wall_clock64();
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B1u + (unsigned)t0;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 0x85EBCA77u + 0x165667B1u;
        const size_t w = ((size_t)((x >> 7) & mask_lines)) * 32 + (threadIdx.x & 31);
        if (MODE == 0) buf[w] = x;
        if (MODE == 1) atomicMax(&buf[w], x);
        if (MODE == 2) reinterpret_cast<uint4 *>(buf)[w / 4] = make_uint4(x, x, x, x);
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
template <class KU> double pair2_us(KU ku, unsigned *big, int per_thread, unsigned *d, hipStream_t st, long long ticks, int n) {
    const unsigned mask = (1u << 23) - 1;  // 8 M lines = 1 GB
    for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(ku, dim3(672), dim3(256), 0, st, big, mask, per_thread, ticks); hipLaunchKernelGGL(k_s<4>, dim3(32), dim3(1024), 0, st, d, ticks); }
    CK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(ku, dim3(672), dim3(256), 0, st, big, mask, per_thread, ticks); hipLaunchKernelGGL(k_s<4>, dim3(32), dim3(1024), 0, st, d, ticks); }
    CK(hipStreamSynchronize(st));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
    unsigned *d;
    CK(hipMalloc(&d, 64));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const long long ticks = 500;  // 5 us of "work" per kernel: a pair cannot take less than 10 us
    const int n = 3000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_s<40960>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    printf("pairs (U: 672 x 256 threads, then S), 5 us of spinning per kernel, us per pair (floor 10.0):\n");
    printf("  S = 32 x 1024 threads, 40 KB static + 8 KB dynamic LDS : %.2f\n", pair_us(k_s<40960>, 32, 1024, 8192, 672, d, st, ticks, n));
    printf("  S = 32 x 1024 threads, 4 B static LDS                  : %.2f\n", pair_us(k_s<4>, 32, 1024, 0, 672, d, st, ticks, n));
    printf("  S = 32 x 512 threads, 40 KB + 8 KB                     : %.2f\n", pair_us(k_s<40960>, 32, 512, 8192, 672, d, st, ticks, n));
    printf("  S = 128 x 256 threads, 4 B                             : %.2f\n", pair_us(k_s<4>, 128, 256, 0, 672, d, st, ticks, n));
    printf("  S = 672 x 256 threads, 4 B (U, U)                      : %.2f\n", pair_us(k_s<4>, 672, 256, 0, 672, d, st, ticks, n));
    printf("  U = 84 x 256 threads; S = 4 x 1024 threads, 40 + 8 KB  : %.2f   (one chain)\n", pair_us(k_s<40960>, 4, 1024, 8192, 84, d, st, ticks, n));
    printf("  U = 84 x 256 threads; S = 4 x 256 threads, 4 B         : %.2f   (one chain)\n", pair_us(k_s<4>, 4, 256, 0, 84, d, st, ticks, n));
    printf("  U = 2688 x 256 threads; S = 32 x 1024, 40 + 8 KB       : %.2f   (four times the update blocks)\n", pair_us(k_s<40960>, 32, 1024, 8192, 2688, d, st, ticks, n));
    unsigned *big;
    CK(hipMalloc(&big, 1ull << 30));
    CK(hipMemset(big, 0, 1ull << 30));
    printf("U2 (672 x 256 threads, every thread writes to random lines of 1 GB, then spins to 5 us) followed by S (32 x 1024): us per pair\n");
    for (int per : {0, 1, 4, 16}) {
        printf("  %2d writes per thread (%6d lines per launch):  4-byte stores %.2f   atomicMax (no return) %.2f   16-byte stores %.2f\n", per, per * 672 * 256,
               pair2_us(k_u2<0>, big, per, d, st, ticks, n), pair2_us(k_u2<1>, big, per, d, st, ticks, n), pair2_us(k_u2<2>, big, per, d, st, ticks, n));
    }
    return 0;
}
