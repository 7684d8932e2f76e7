// TEST INFRASTRUCTURE -- does the ThreadSanitizer build of the emulated device see what it is supposed to see?
// Small kernels with and without the races the detector is meant for; built with -fsanitize=thread and run by
// tests/test_emulated_device.py, which expects a report for every racy case and none for the clean ones.
//   usage: race_selftest <case>     cases: clean_barrier clean_wave clean_atomic clean_lds_blocks clean_launches race_blocks race_blocks_after_barrier race_waves race_lanes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void k_blocks_same_word(int *p) {  // every block writes the same word: racy between blocks
    if (threadIdx.x == 0) p[0] = (int)blockIdx.x;
}
__global__ void k_blocks_after_barrier(int *p) {  // the barriers of different blocks must not order the blocks
    __shared__ int box;
    if (threadIdx.x == 0) box = 1;
    __syncthreads();
    if (threadIdx.x == 65) p[0] = box + (int)blockIdx.x;
}
__global__ void k_blocks_atomic(int *p) {
    if (threadIdx.x == 0) atomicAdd(&p[0], 1);
}
__global__ void k_waves(int *p, int use_barrier) {  // wave 0 writes, wave 1 reads: needs the block barrier
    __shared__ int box;
    if (threadIdx.x == 0) box = 42;
    if (use_barrier) __syncthreads();
    if (threadIdx.x == 64) p[0] = box;
}
__global__ void k_lanes(int *p, int use_fence) {  // lane 0 writes, lane 1 reads: needs a wave-level rendezvous
    __shared__ int box;
    if (threadIdx.x == 0) box = 7;
    if (use_fence) __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 1) p[0] = box;
}
__global__ void k_lds_per_block(int *p) {  // LDS is per block: the same static and dynamic bytes in every block, no conflict
    __shared__ int acc;
    int *dyn = reinterpret_cast<int *>(::hipemu::dyn_shared());
    if (threadIdx.x == 0) acc = 0;
    if (threadIdx.x < 7) dyn[threadIdx.x] = 0;
    __syncthreads();
    atomicAdd(&acc, 1);
    atomicAdd(&dyn[threadIdx.x % 7], 1);
    __syncthreads();
    if (threadIdx.x == 0) p[blockIdx.x] = acc + dyn[0];
}
__global__ void k_write(int *p) { p[threadIdx.x] = (int)threadIdx.x; }
__global__ void k_read(const int *p, int *q) { q[threadIdx.x] = p[(threadIdx.x + 1) % 128]; }

int main(int argc, char **argv) {
    const char *what = argc > 1 ? argv[1] : "";
    int *p, *q;
    hipMalloc(&p, 4096);
    hipMalloc(&q, 4096);
    std::memset(p, 0, 4096);
    if (!std::strcmp(what, "race_blocks")) hipLaunchKernelGGL(k_blocks_same_word, dim3(4), dim3(64), 0, nullptr, p);
    else if (!std::strcmp(what, "race_blocks_after_barrier")) hipLaunchKernelGGL(k_blocks_after_barrier, dim3(4), dim3(128), 0, nullptr, p);
    else if (!std::strcmp(what, "clean_atomic")) hipLaunchKernelGGL(k_blocks_atomic, dim3(4), dim3(64), 0, nullptr, p);
    else if (!std::strcmp(what, "race_waves")) hipLaunchKernelGGL(k_waves, dim3(1), dim3(128), 0, nullptr, p, 0);
    else if (!std::strcmp(what, "clean_barrier")) hipLaunchKernelGGL(k_waves, dim3(1), dim3(128), 0, nullptr, p, 1);
    else if (!std::strcmp(what, "race_lanes")) hipLaunchKernelGGL(k_lanes, dim3(1), dim3(64), 0, nullptr, p, 0);
    else if (!std::strcmp(what, "clean_wave")) hipLaunchKernelGGL(k_lanes, dim3(1), dim3(64), 0, nullptr, p, 1);
    else if (!std::strcmp(what, "clean_lds_blocks")) hipLaunchKernelGGL(k_lds_per_block, dim3(5), dim3(256), 28, nullptr, p);  // 28: not a whole number of 8-byte cells
    else if (!std::strcmp(what, "clean_launches")) {  // a kernel boundary orders everything
        hipLaunchKernelGGL(k_write, dim3(1), dim3(128), 0, nullptr, p);
        hipLaunchKernelGGL(k_read, dim3(1), dim3(128), 0, nullptr, p, q);
    } else {
        std::fprintf(stderr, "unknown case\n");
        return 2;
    }
    std::printf("%s done %d\n", what, p[0] + q[0]);
    return 0;
}
