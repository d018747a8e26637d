// Launch-bound regime (SURVEY section 8e strong scaling: 1M filters over 8 GPUs = 131k per GPU; smaller batches more so): T dependent
// single-step updates of one batch as (a) T kb_update_dev calls, (b) the same T calls captured ONCE into a hipGraph on the handle's
// stream and replayed, (c) kb_update_steps_dev (one time-fused kernel, where it exists).  Per-step time for several batch sizes.
// hipcc -std=c++17 -O2 -Iinclude scripts/diag_graph.cpp -Lgokalman_amd -lgokalman_amd -Wl,-rpath,$PWD/gokalman_amd -o /tmp/diag_graph
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#include "gokalman_amd.h"

#define CK(x) do { if ((x) != 0) { std::fprintf(stderr, "%s failed: %s\n", #x, kb_last_error()); return 3; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 4; } } while (0)

int main() {
    const int n = 6, p = 3, T = 64;
    for (long N : {64L, 4096L, 16384L, 131072L, 1048576L}) {
        kb_batch *b = nullptr;
        CK(kb_create(&b, KB_VANILLA, n, p, 0, N, KB_F64, 0, 0));
        std::vector<double> x0(n, 0.1), P0(n * n, 0.0), F(n * n, 0.0), H(p * n, 0.0), Q(n * n, 0.0), R(p * p, 0.0);
        for (int i = 0; i < n; i++) { P0[i * n + i] = 2.0; F[i * n + i] = 1.0; Q[i * n + i] = 1e-3; if (i + 3 < n) F[i * n + i + 3] = 0.1; }
        for (int i = 0; i < p; i++) { H[i * n + i] = 1.0; R[i * p + i] = 0.05; }
        CK(kb_set(b, KB_X, x0.data(), 1, 1, 0)); CK(kb_set(b, KB_P, P0.data(), 1, 1, 0)); CK(kb_set(b, KB_F, F.data(), 1, 1, 0));
        CK(kb_set(b, KB_H, H.data(), 1, 1, p)); CK(kb_set(b, KB_Q, Q.data(), 1, 1, 0)); CK(kb_set(b, KB_R, R.data(), 1, 1, p));
        CK(kb_init(b));
        const long ld = (N + 63) / 64 * 64;
        double *dy = nullptr;
        HK(hipMalloc(&dy, (size_t)T * p * ld * sizeof(double)));
        HK(hipMemset(dy, 0, (size_t)T * p * ld * sizeof(double)));
        hipStream_t s = (hipStream_t)kb_stream(b);
        auto loop = [&]() { for (int t = 0; t < T; t++) kb_update_dev(b, dy + (size_t)t * p * ld, ld, nullptr, 0); return 0; };
        auto timeit = [&](auto &&fn, int reps) {
            fn(); kb_synchronize(b);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) fn();
            kb_synchronize(b);
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / T;
        };
        const int reps = N >= 1048576 ? 4 : 40;
        const double plain = timeit(loop, reps);
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        HK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        loop();
        HK(hipStreamEndCapture(s, &g));
        HK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double graph = timeit([&]() { hipGraphLaunch(ge, s); return 0; }, reps);
        const double fused = timeit([&]() { kb_update_steps_dev(b, dy, ld, nullptr, 0, T); return 0; }, reps);
        std::printf("N = %8ld: %7.2f us per step with %d kb_update_dev calls, %7.2f as one hipGraph of %d kernel nodes, %7.2f time-fused (kb_update_steps_dev)\n",
                    N, plain, T, graph, T, fused);
        hipGraphExecDestroy(ge); hipGraphDestroy(g); hipFree(dy); kb_destroy(b);
    }
    return 0;
}
