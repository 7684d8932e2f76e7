// Which HIP streams make progress SIDE BY SIDE on MI355X?  Streams are mapped onto GPU_MAX_HW_QUEUES HSA queues (default 4) and those onto the
// pipes of the command processor; chains of dependent kernels (barrier bit set) on two queues of one pipe take turns instead of overlapping.
//   mode "scan":   Q streams, each a chain of K kernels spinning `us` microseconds, all queued eagerly; wall time per kernel of a chain
//   mode "matrix": N streams created the way the engine creates them (one after the other, non-blocking); every PAIR timed: 1 = side by side,
//                  2 = one after the other; then the largest set of streams that all overlap pairwise, timed together
//   hipcc --offload-arch=gfx950 -O3 -o queue_probe queue_probe.hip && GPU_MAX_HW_QUEUES=8 ./queue_probe matrix 12
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

__global__ void spin(long long ticks, unsigned int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0xFFFFFFu) *sink = 1;
}

static double run_set(const std::vector<hipStream_t> &st, int K, long long ticks, int blocks) {  // us per kernel of a chain
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; ++k)
        for (auto s : st) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, ticks, nullptr);
    CK(hipDeviceSynchronize());
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
}

int main(int argc, char **argv) {
    int rate = 0;
    CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));  // kHz
    const char *env = getenv("GPU_MAX_HW_QUEUES");
    if (argc > 1 && !strcmp(argv[1], "matrix")) {
        const int N = argc > 2 ? atoi(argv[2]) : 9, K = 60;
        const double us = 25.0;
        const long long ticks = (long long)(us * rate / 1000.0);
        std::vector<hipStream_t> st(N);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto s : st) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, s, ticks, nullptr);
        printf("GPU_MAX_HW_QUEUES=%s: %d streams in creation order; pair (i, j): 1 = side by side, 2 = taking turns (chains of %d kernels of %.0f us)\n", env ? env : "(default)", N, K, us);
        std::vector<std::vector<int>> par(N, std::vector<int>(N, 0));
        for (int i = 0; i < N; ++i) {
            printf("  %2d:", i);
            for (int j = 0; j < N; ++j) {
                if (j <= i) { printf("  ."); continue; }
                const double t = run_set({st[i], st[j]}, K, ticks, 32);
                par[i][j] = par[j][i] = t < 1.5 * us;
                printf("  %d", t < 1.5 * us ? 1 : 2);
            }
            printf("\n");
        }
        std::vector<int> best;  // greedy clique from every start
        for (int s0 = 0; s0 < N; ++s0) {
            std::vector<int> c{s0};
            for (int j = 0; j < N; ++j) {
                bool ok = j != s0;
                for (int m : c) ok = ok && par[m][j];
                if (ok) c.push_back(j);
            }
            if (c.size() > best.size()) best = c;
        }
        printf("  largest set that overlaps pairwise:");
        std::vector<hipStream_t> cs;
        for (int m : best) { printf(" %d", m); cs.push_back(st[m]); }
        for (int blocks : {32, 512}) {
            const double t = run_set(cs, 200, ticks, blocks);
            printf("\n  all %zu together, %d blocks: %.2f us per kernel of a chain -> %.2f in flight", cs.size(), blocks, t, cs.size() * us / t);
        }
        std::vector<hipStream_t> first4(st.begin(), st.begin() + (N < 4 ? N : 4));
        const double t4 = run_set(first4, 200, ticks, 32);
        printf("\n  streams 0-3 together (the engine's choice): %.2f us -> %.2f in flight\n", t4, first4.size() * us / t4);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "masked")) {  // four CU-masked streams (each gets an HSA queue of its own) created after `pre` ordinary ones were used
        const int pre = argc > 2 ? atoi(argv[2]) : 9;
        const double us = 25.0;
        const long long ticks = (long long)(us * rate / 1000.0);
        std::vector<hipStream_t> st(pre), ms(4);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto s : st) hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, s, ticks, nullptr);
        CK(hipDeviceSynchronize());
        const bool quarter = argc > 3 && !strcmp(argv[3], "quarter");
        for (int l = 0; l < 4; ++l) {
            uint32_t m[8];
            for (int w = 0; w < 8; ++w) m[w] = quarter ? (w / 2 == l ? 0xFFFFFFFFu : 0u) : 0xFFFFFFFFu;
            CK(hipExtStreamCreateWithCUMask(&ms[l], 8, m));
        }
        printf("GPU_MAX_HW_QUEUES=%s: %d ordinary streams used, then four CU-masked streams (%s mask)\n", env ? env : "(default)", pre, quarter ? "64 CUs each" : "full");
        for (int i = 0; i < 4; ++i) {
            printf("  masked %d against masked:", i);
            for (int j = 0; j < 4; ++j) printf("  %s", j <= i ? "." : (run_set({ms[i], ms[j]}, 60, ticks, 32) < 1.5 * us ? "1" : "2"));
            printf("   against ordinary 0-%d:", pre - 1);
            for (int j = 0; j < pre; ++j) printf("  %s", run_set({ms[i], st[j]}, 60, ticks, 32) < 1.5 * us ? "1" : "2");
            printf("\n");
        }
        for (int blocks : {32, 512}) {
            const double t = run_set(ms, 200, ticks, blocks);
            printf("  the four masked streams together, %d blocks: %.2f us per kernel of a chain -> %.2f in flight\n", blocks, t, 4 * us / t);
        }
        const double t1 = run_set({ms[0]}, 200, ticks, 32);
        printf("  one masked stream alone: %.2f us per kernel\n", t1);
        return 0;
    }
    const int K = argc > 1 ? atoi(argv[1]) : 2000;
    const double us = argc > 2 ? atof(argv[2]) : 10.0;
    const long long ticks = (long long)(us * rate / 1000.0);
    printf("GPU_MAX_HW_QUEUES=%s, chains of %d kernels spinning %.1f us\n", env ? env : "(default)", K, us);
    for (int blocks : {32, 512}) {
        for (int Q : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
            std::vector<hipStream_t> st(Q);
            for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            for (auto &s : st) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, ticks, nullptr);  // warm the queues
            const double t = run_set(st, K, ticks, blocks);
            printf("  %4d blocks x 256, %2d streams: %7.2f us per kernel of a chain  -> %.2f kernels in flight on average\n", blocks, Q, t, Q * us / t);
            for (auto &s : st) CK(hipStreamDestroy(s));
        }
    }
    return 0;
}
