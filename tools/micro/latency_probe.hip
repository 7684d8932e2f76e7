// Dependent-load latency of random 128-byte lines vs footprint and vs the number of chasing waves (MI355X).
// Decides whether the greedy loop's round trips are stretched by address translation / footprint or by queueing.
//   hipcc --offload-arch=gfx950 -O3 -o latency_probe latency_probe.hip && ./latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

// every wave chases its own pseudo-random sequence of lines: next = hash(value read); all 64 lanes read one line
__global__ void chase(const unsigned long long *buf, unsigned long long n_lines, int steps, unsigned long long *out, long long *cycles) {
    const int lane = threadIdx.x & 63;
    unsigned long long gw = (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
    unsigned long long x = gw * 0x9E3779B97F4A7C15ull + 12345;
    long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 29;
        unsigned long long line = x % n_lines;
        unsigned long long v = buf[line * 16 + (lane & 15)];
        x += v;  // dependent
    }
    long long t1 = wall_clock64();
    if (lane == 0) { out[gw] = x; cycles[gw] = t1 - t0; }
}
// the same chase with a cheap index (32-bit multiply + mask instead of 64-bit multiplies and a modulo: the arithmetic of `chase` is part of its
// figure) over SMALL footprints: is a dependent line load that hits the XCD's 4 MB L2 or the 256 MB Infinity Cache faster than one that goes to HBM?
// (round 6: decides whether compacting the pair table of an old chain can pay)
__global__ void chase_pow2(const unsigned long long *buf, unsigned int line_mask, int steps, unsigned long long *out, long long *cycles) {
    const int lane = threadIdx.x & 63;
    const unsigned int gw = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    unsigned int x = gw * 0x9E3779B1u + 12345u;
    long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
        x = x * 0x85EBCA77u + 0x165667B1u;
        const unsigned int line = (x >> 9) & line_mask;
        const unsigned long long v = buf[(size_t)line * 16 + (lane & 15)];
        x += (unsigned int)v;  // dependent
    }
    long long t1 = wall_clock64();
    if (lane == 0) { out[gw] = x; cycles[gw] = t1 - t0; }
}
int main(int argc, char **argv) {
    const size_t GB = 1ull << 30;
    if (argc > 1 && argv[1][0] == 's') {  // ./latency_probe small
        const size_t MB = 1ull << 20;
        unsigned long long *buf;
        CK(hipMalloc(&buf, 1024 * MB));
        CK(hipMemset(buf, 0, 1024 * MB));
        unsigned long long *out; long long *cyc;
        CK(hipMalloc(&out, 8 << 20)); CK(hipMalloc(&cyc, 8 << 20));
        int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));
        printf("wall clock %d kHz; dependent random 128-byte line loads, cheap index arithmetic, all 256 CUs chasing inside the SAME footprint\n", rate);
        const int steps = 4000;
        for (size_t mb : {1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024}) {
            const unsigned int mask = (unsigned int)(mb * MB / 128 - 1);
            for (int waves_per_cu : {1, 4, 16}) {
                const int blocks = 256 * waves_per_cu / 4 < 1 ? 1 : 256 * waves_per_cu / 4;
                for (int rep = 0; rep < 2; ++rep) {  // (the first pass warms the caches)
                    hipLaunchKernelGGL(chase_pow2, dim3(blocks), dim3(256), 0, 0, buf, mask, steps, out, cyc);
                    CK(hipDeviceSynchronize());
                }
                std::vector<long long> h(blocks * 4);
                CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
                double s = 0; for (auto v : h) s += v;
                const double ns = s / h.size() / steps * 1e6 / rate;
                printf("footprint %5zu MB  waves/CU %2d : %7.1f ns per dependent random line load\n", mb, waves_per_cu, ns);
            }
        }
        return 0;
    }
    size_t max_bytes = 48 * GB;
    unsigned long long *buf;
    CK(hipMalloc(&buf, max_bytes));
    CK(hipMemset(buf, 0, max_bytes));
    unsigned long long *out; long long *cyc;
    CK(hipMalloc(&out, 8 << 20)); CK(hipMalloc(&cyc, 8 << 20));
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));  // kHz
    printf("wall clock %d kHz\n", rate);
    const int steps = 2000;
    for (double gb : {0.25, 1.0, 4.0, 8.0, 16.0, 32.0, 48.0}) {
        unsigned long long n_lines = (unsigned long long)(gb * GB) / 128;
        for (int waves_per_cu : {1, 8, 24}) {
            int blocks = 256 * waves_per_cu / 4;
            if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 0, 0, buf, n_lines, steps, out, cyc);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(blocks * 4);
            CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            double s = 0; for (auto v : h) s += v;
            double ns = s / h.size() / steps * 1e6 / rate;
            printf("footprint %6.2f GB  waves/CU %2d : %7.1f ns per dependent random line load  (%.0f GB/s aggregate)\n", gb, waves_per_cu, ns, (double)h.size() * 128 / ns);
        }
    }
    return 0;
}
