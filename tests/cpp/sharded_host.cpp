// gokalman::ShardedBatch (include/gokalman_amd.hpp over kb_sharded_*): one process driving several shards, each with its own
// handle, host thread and stream.  On a one-GPU box both shards live on device 0 (the statistics reduction is then a host sum;
// ncclAllReduce needs distinct devices).  Compared with an unsharded gokalman::Batch of the same filters.
#include <cmath>
#include <cstdio>

#include "gokalman_amd.hpp"

using namespace gokalman;

int main() {
    try {
        const int n = 4, p = 2;
        const int64_t N = 1500;
        Matrix F(n, n, {1, 0.1, 0, 7.726e-2, 4.015e-7, 1, 0, 1.545, -2.319e-16, -1.732e-9, 1, 0.1, -6.956e-15, -3.465e-8, 0, 1});
        Matrix G(n, 2, {5e-3, 3.85e-7, 0.1, 1.157e-5, -5.775e-11, 7.487e-7, 1.732e-9, 1.498e-5});
        Matrix H(p, n, {1, 0, 0, 0, 0, 0, 1, 0});
        Matrix Q(n, n, {6.669e-16, 1.001e-14, 3.823e-19, 5.150e-18, 1.001e-14, 2.002e-13, 1.030e-17, 1.545e-16,
                        3.862e-19, 1.030e-17, 6.667e-19, 1.000e-17, 5.150e-18, 1.545e-16, 1.000e-17, 2.000e-16});
        Matrix R(p, p, {2e-2, 0, 0, 2e-4});
        Matrix P0(n, n, {5, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0.01, 0, 0, 0, 0, 1e-5});
        // per-filter initial states and measurements
        Matrix x0(n, 1);
        x0.data.assign((size_t)N * n, 0.0);
        std::vector<double> y((size_t)N * p), u((size_t)N * 2, 0.0);
        for (int64_t i = 0; i < N; i++) {
            for (int e = 0; e < n; e++) x0.data[(size_t)i * n + e] = std::sin(0.37 * (double)(i * n + e));
            for (int e = 0; e < p; e++) y[(size_t)i * p + e] = std::cos(0.11 * (double)(i * p + e));
        }
        const Noise quiet = NewNoiseless(Q, R);
        ShardedBatch sh(KB_VANILLA, x0, P0, F, G, H, quiet, N, {0, 0});
        auto one = NewVanilla(x0, P0, F, G, H, quiet, N, 0, 0u).first;
        Vector Y(p, 1); Y.data = y;
        Vector U(2, 1); U.data = u;
        for (int t = 0; t < 5; t++) {
            sh.Update(Y, U);
            (void)one->Update(Y, U);
        }
        std::printf("shards %d\n", sh.Shards());
        std::printf("bit_equal_state %d\n", sh.State().data == one->batch()->get(KB_STATE, n, 1).data ? 1 : 0);
        std::printf("bit_equal_covariance %d\n", sh.Covariance().data == one->batch()->get(KB_COVAR, n, n).data ? 1 : 0);
        // Monte-Carlo + chi-square: the sharded ensemble against one batch holding every run (same seed => same runs)
        const Noise awgn = NewAWGN(Q, R, 99);
        const int steps = 25;
        const std::vector<Vector> zero{NewVector(2)};
        Vector x00 = NewVector(n, {2, 0.5, 0, 0});
        ShardedBatch truth(KB_VANILLA_PREDICT, x00, P0, F, G, H, awgn, N, {0, 0});
        ShardedBatch kfs(KB_VANILLA, x00, P0, F, G, H, quiet, N, {0, 0});
        const auto st = truth.MonteCarlo(steps, zero);
        const auto chi = truth.ChiSquare(kfs, steps, zero, true);
        auto mcb = NewPurePredictorVanilla(x00, P0, F, G, H, awgn, N, 0, 0u).first;
        auto kfb = NewVanilla(x00, P0, F, G, H, quiet, N, 0, 0u).first;
        const MonteCarloRuns runs = NewMonteCarloRuns(N, steps, p, zero, *mcb, 0);
        const auto chi1 = NewChiSquare(*kfb, runs, zero, true, true);
        bool mc_ok = true, chi_ok = true;
        for (int t = 0; t < steps; t++) {
            for (int i = 0; i < n; i++) {
                const double a = st.mean[(size_t)t * n + i], b = runs.Mean(t)[(size_t)i];
                const double c = st.stddev[(size_t)t * n + i], d = runs.StdDev(t)[(size_t)i];
                mc_ok = mc_ok && std::fabs(a - b) <= 1e-12 * std::fabs(b) + 1e-15 && std::fabs(c - d) <= 1e-9 * std::fabs(d);
            }
            chi_ok = chi_ok && std::fabs(chi.first[(size_t)t] - chi1.first[(size_t)t]) <= 1e-10 * std::fabs(chi1.first[(size_t)t]) &&
                     std::fabs(chi.second[(size_t)t] - chi1.second[(size_t)t]) <= 1e-10 * std::fabs(chi1.second[(size_t)t]);
        }
        std::printf("mc_matches_single_batch %d\nchi_matches_single_batch %d\nused_rccl %d\n", mc_ok ? 1 : 0, chi_ok ? 1 : 0, st.usedRccl ? 1 : 0);
    } catch (const Error &e) {
        std::fprintf(stderr, "gokalman error %d: %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}
