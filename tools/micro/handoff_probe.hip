// What does a device-side hand-off between workgroups cost on MI355X, compared with a kernel boundary?
// Input for the "persistent, decoupled chains" design: the greedy loop today pays two kernel
// boundaries per step; a resident team per chain would pay two team barriers + the coherence traffic instead.
//   (1) dependent kernels back to back on one stream: time per kernel boundary (grid of 64 and of 2560 blocks)
//   (2) persistent kernel, T teams x B blocks, R rounds of a team barrier through a device-scope atomic counter
//   (3) producer block -> consumer blocks: payload written with plain stores, released with an agent-scope fence +
//       flag, acquired and CHECKED by the consumers (stale data = the per-XCD L2s were not made coherent), acknowledged
// Every spin is bounded (a failed barrier sets an error flag, the kernel drains) so that nothing can hang the box.
//   hipcc --offload-arch=gfx950 -O3 -o handoff_probe handoff_probe.hip && ./handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

constexpr int MAX_POLLS = 1 << 18;

__global__ void k_empty(unsigned int *sink) {
    if (sink == nullptr) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] += 1;
}

// wait until *p >= target (agent scope; relaxed polls, then an acquire fence); false = gave up
__device__ bool spin_until(unsigned int *p, unsigned int target, unsigned int *err) {
    for (int i = 0; i < MAX_POLLS; ++i) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {  // polls bypass the caches, no invalidation
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                              // one acquire when the wait is over
            return true;
        }
        if ((i & 255) == 255 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// (2) team barrier: blocks [team * B, (team + 1) * B) meet R times
__global__ void __launch_bounds__(256) k_team_barrier(unsigned int *cnt, int B, int R, unsigned int *err, long long *ticks) {
    const int team = blockIdx.x / B;
    unsigned int *c = cnt + team * 32;  // one 128-byte line per team
    __shared__ int ok;
    const long long t0 = wall_clock64();
    for (int r = 0; r < R; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            ok = spin_until(c, (unsigned)(r + 1) * B, err);
        }
        __syncthreads();
        if (!ok) break;
    }
    if (threadIdx.x == 0 && blockIdx.x % B == 0) ticks[team] = wall_clock64() - t0;
}

// (3) producer (block 0) -> consumers (blocks 1..): payload words must equal the round number when the flag says so
__global__ void __launch_bounds__(256) k_handoff(unsigned int *payload, int words, unsigned int *flag, unsigned int *acks, int R, unsigned int *err,
                                                 unsigned int *stale, long long *ticks) {
    const int consumers = gridDim.x - 1;
    __shared__ int ok;
    const long long t0 = wall_clock64();
    unsigned int bad = 0;
    for (int r = 1; r <= R; ++r) {
        if (blockIdx.x == 0) {
            for (int i = threadIdx.x; i < words; i += blockDim.x) payload[i] = (unsigned)r;  // plain stores
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_store(flag, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // release: write back this XCD's L2
                ok = spin_until(acks, (unsigned)r * consumers, err);
            }
            __syncthreads();
            if (!ok) break;
        } else {
            if (threadIdx.x == 0) ok = spin_until(flag, (unsigned)r, err);  // acquire: invalidate this XCD's L2 / the CU's L1
            __syncthreads();
            if (!ok) break;
            for (int i = threadIdx.x; i < words; i += blockDim.x) bad += payload[i] != (unsigned)r;  // plain loads
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(acks, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bad) atomicAdd(stale, bad);
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = wall_clock64() - t0;
}

int main() {
    int rate = 0;
    CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));  // kHz
    const double us_per_tick = 1e3 / rate;
    unsigned int *buf;
    long long *ticks;
    CK(hipMalloc(&buf, 64 << 20));
    CK(hipMalloc(&ticks, 1 << 20));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    // (1) kernel boundary
    for (int blocks : {1, 64, 2560}) {
        const int K = 2000;
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st, buf);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st, buf);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("kernel boundary: %4d blocks x 256 threads, %d dependent launches: %.2f us per launch\n", blocks, K, 1e3 * ms / K);
    }

    // (2) team barriers.  All blocks must be resident together: <= 1024 blocks of 256 threads (4 per CU)
    const int cfg[][2] = {{64, 2}, {64, 4}, {64, 8}, {64, 16}, {21, 8}, {21, 48}, {3, 256}, {1, 256}, {1, 1024}};
    for (auto &tb : cfg) {
        const int T = tb[0], B = tb[1], R = 2000;
        CK(hipMemsetAsync(buf, 0, 1 << 20, st));
        unsigned int *cnt = buf, *err = buf + (1 << 16);
        hipLaunchKernelGGL(k_team_barrier, dim3(T * B), dim3(256), 0, st, cnt, B, R, err, ticks);
        CK(hipStreamSynchronize(st));
        std::vector<long long> h(T);
        unsigned int herr = 0;
        CK(hipMemcpy(h.data(), ticks, T * sizeof(long long), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        long long mx = 0;
        for (long long v : h) mx = v > mx ? v : mx;
        printf("team barrier: %3d teams x %4d blocks, %d rounds: %.2f us per barrier%s\n", T, B, R, mx * us_per_tick / R, herr ? "  [GAVE UP: not co-resident?]" : "");
    }

    // (3) producer -> consumers with payload check
    for (int consumers : {1, 7, 31}) {
        for (int bytes : {256, 4096, 65536}) {
            const int R = 1000, words = bytes / 4;
            CK(hipMemsetAsync(buf, 0, 4 << 20, st));
            unsigned int *payload = buf + (1 << 18), *flag = buf, *acks = buf + 32, *err = buf + 64, *stale = buf + 96;
            hipLaunchKernelGGL(k_handoff, dim3(1 + consumers), dim3(256), 0, st, payload, words, flag, acks, R, err, stale, ticks);
            CK(hipStreamSynchronize(st));
            long long t = 0;
            unsigned int herr = 0, hstale = 0;
            CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&hstale, stale, 4, hipMemcpyDeviceToHost));
            printf("hand-off: 1 producer -> %2d consumer blocks, %6d B payload, %d round trips: %.2f us per round trip, stale words %u%s\n", consumers, bytes, R,
                   t * us_per_tick / R, hstale, herr ? "  [GAVE UP]" : "");
        }
    }
    return 0;
}
