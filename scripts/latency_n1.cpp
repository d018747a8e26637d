// Per-call latency of the drop-in path (one filter, host vectors) through the C ABI, without any binding overhead:
//   kb_update alone; kb_update + kb_get_estimate(state, covariance, status); kb_update + kb_get_estimate(every member);
//   kb_update + the round-1 pattern of separate kb_get calls.
// hipcc -std=c++17 -O2 -Iinclude scripts/latency_n1.cpp -Lgokalman_amd -lgokalman_amd -Wl,-rpath,$PWD/gokalman_amd -o scripts/latency_n1
#include <chrono>
#include <cstdio>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "gokalman_amd.h"

int main() {
    const int n = 6, p = 3;
    kb_batch *b = nullptr;
    if (kb_create(&b, KB_VANILLA, n, p, 0, 1, KB_F64, 0, KB_FLAG_FULL_ESTIMATE)) { std::fprintf(stderr, "%s\n", kb_last_error()); return 3; }
    std::vector<double> x0(n, 0.1), P0(n * n, 0.0), F(n * n, 0.0), H(p * n, 0.0), Q(n * n, 0.0), R(p * p, 0.0), y(p, 0.3);
    for (int i = 0; i < n; i++) { P0[i * n + i] = 2.0; F[i * n + i] = 1.0; Q[i * n + i] = 1e-3; if (i + 3 < n) F[i * n + i + 3] = 0.1; }
    for (int i = 0; i < p; i++) { H[i * n + i] = 1.0; R[i * p + i] = 0.05; }
    kb_set(b, KB_X, x0.data(), 1, 1, 0); kb_set(b, KB_P, P0.data(), 1, 1, 0); kb_set(b, KB_F, F.data(), 1, 1, 0);
    kb_set(b, KB_H, H.data(), 1, 1, p); kb_set(b, KB_Q, Q.data(), 1, 1, 0); kb_set(b, KB_R, R.data(), 1, 1, p);
    if (kb_init(b)) { std::fprintf(stderr, "%s\n", kb_last_error()); return 3; }
    std::vector<double> xs(n), Pc(n * n), Pp(n * n), K(n * p), in(p), yh(p);
    uint32_t st = 0;
    auto timeit = [&](const char *name, auto &&fn) {
        for (int i = 0; i < 200; i++) fn();
        const int reps = 5000;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) fn();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("%-70s %7.1f us per call\n", name, us);
    };
    {   // where the time goes: enqueue cost of the step alone (device-resident y, no wait) and the snapshot alone
        void *dy = nullptr;
        hipMalloc(&dy, 64 * p * sizeof(double));
        hipMemset(dy, 0, 64 * p * sizeof(double));
        timeit("kb_update_dev x 1 (enqueue only; one kb_synchronize per 5000)", [&] { kb_update_dev(b, dy, 64, nullptr, 0); });
        kb_synchronize(b);
        timeit("kb_update_dev + kb_synchronize", [&] { kb_update_dev(b, dy, 64, nullptr, 0); kb_synchronize(b); });
        timeit("kb_get_estimate(all six members, status) alone", [&] {
            kb_estimate_view v{}; v.state = xs.data(); v.covariance = Pc.data(); v.pred_covariance = Pp.data(); v.gain = K.data();
            v.innovation = in.data(); v.measurement = yh.data(); v.status = &st; v.clear_status = 1;
            kb_get_estimate(b, 0, 1, &v);
        });
        hipFree(dy);
    }
    timeit("kb_update", [&] { kb_update(b, y.data(), p, nullptr, 0); });
    timeit("kb_update + kb_get_estimate(state, covariance, status)", [&] {
        kb_update(b, y.data(), p, nullptr, 0);
        kb_estimate_view v{}; v.state = xs.data(); v.covariance = Pc.data(); v.status = &st; v.clear_status = 1;
        kb_get_estimate(b, 0, 1, &v);
    });
    timeit("kb_update + kb_get_estimate(all six members, status)", [&] {
        kb_update(b, y.data(), p, nullptr, 0);
        kb_estimate_view v{}; v.state = xs.data(); v.covariance = Pc.data(); v.pred_covariance = Pp.data(); v.gain = K.data();
        v.innovation = in.data(); v.measurement = yh.data(); v.status = &st; v.clear_status = 1;
        kb_get_estimate(b, 0, 1, &v);
    });
    timeit("kb_update_estimate(all six members, status): ONE synchronisation", [&] {
        kb_estimate_view v{}; v.state = xs.data(); v.covariance = Pc.data(); v.pred_covariance = Pp.data(); v.gain = K.data();
        v.innovation = in.data(); v.measurement = yh.data(); v.status = &st; v.clear_status = 1;
        kb_update_estimate(b, y.data(), p, nullptr, 0, 0, 1, &v);
    });
    timeit("kb_update + kb_get(STATE) + kb_get(COVAR) + kb_get_status  (round 1)", [&] {
        kb_update(b, y.data(), p, nullptr, 0);
        kb_get(b, KB_STATE, xs.data(), 0, 1); kb_get(b, KB_COVAR, Pc.data(), 0, 1); kb_get_status(b, &st, 0, 1);
    });
    timeit("kb_update + six kb_get + kb_get_status  (round 1, full estimate)", [&] {
        kb_update(b, y.data(), p, nullptr, 0);
        kb_get(b, KB_STATE, xs.data(), 0, 1); kb_get(b, KB_COVAR, Pc.data(), 0, 1); kb_get(b, KB_PRED_COVAR, Pp.data(), 0, 1);
        kb_get(b, KB_GAIN, K.data(), 0, 1); kb_get(b, KB_INNOVATION, in.data(), 0, 1); kb_get(b, KB_MEASUREMENT, yh.data(), 0, 1);
        kb_get_status(b, &st, 0, 1);
    });
    kb_destroy(b);
    return 0;
}
