// Estimate value semantics and per-call errors of the C++ host mirror (include/gokalman_amd.hpp) over the C ABI.
//  1. an Estimate held across the next Update still reads ITS step (vanilla.go:216-218: a fresh estimate per Update);
//  2. an Update that fails numerically throws (the reference's `return nil, err`, vanilla.go:164-167), leaves the
//     previous estimate in place, and the next valid Update succeeds;
//  3. the same for an SRIF whose Phi is singular at one step (srif.go:112-114);
//  4. kf.step is not advanced by the failed call (vanilla.go:164-167 returns before :218; srif.go:112-114 before :157), and
//     the SRIF stays prepared (srif.go:158 `kf.locked = true` is not reached);
//  5. NewMonteCarloRuns(samples, steps, rowsH, controls, kf) / runs.Runs[r].Estimates[k] / runs.AsCSV(headers) /
//     NewChiSquare(kf, runs, controls, withNEES, withNIS) with the reference's signatures (montecarlo.go:92, :62-89, chisquare.go:16).
// Prints "name v0 v1 ..." lines; tests/test_cpp_host.py compares them with the oracle.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "gokalman_amd.hpp"

using namespace gokalman;

static void print_vec(const char *name, const Matrix &m) {
    std::printf("%s", name);
    for (double v : m.data) std::printf(" %.17g", v);
    std::printf("\n");
}

int main() {
    try {
        // ---- 1 + 2: Vanilla, 2 states / 1 measurement (examples/robot-sized) ----------------------------------------
        Matrix F(2, 2, {1, 0.1, 0, 1}), G(2, 1), H(1, 2, {1, 0}), Hzero(1, 2, {0, 0});
        const Noise good = NewNoiseless(Matrix(2, 2, {1e-3, 0, 0, 1e-3}), Matrix(1, 1, {0.05}));
        const Noise zeroR = NewNoiseless(Matrix(2, 2, {1e-3, 0, 0, 1e-3}), Matrix(1, 1, {0.0}));
        auto pr = NewVanilla(NewVector(2, {0.5, -0.2}), ScaledIdentity(2, 4.0), F, G, H, good);
        auto kf = pr.first;
        const Estimate est0 = pr.second;
        const Estimate est1 = kf->Update(NewVector(1, {0.7}), NewVector(0));
        const Vector x1_then = est1.State();
        const Matrix P1_then = est1.Covariance();
        const Estimate est2 = kf->Update(NewVector(1, {0.9}), NewVector(0));
        // est1 (and est0) are values: unchanged by the later Updates
        const bool held = est1.State().data == x1_then.data && est1.Covariance().data == P1_then.data &&
                          est1.State().data != est2.State().data && est0.State().data == std::vector<double>({0.5, -0.2});
        std::printf("held_estimate_unchanged %d\n", held ? 1 : 0);
        print_vec("x1", est1.State()); print_vec("P1", est1.Covariance());
        print_vec("x2", est2.State()); print_vec("P2", est2.Covariance());
        print_vec("K2", est2.Gain()); print_vec("innov2", est2.Innovation()); print_vec("Ppred2", est2.PredCovariance());
        // String() of the estimate and of the filter (vanilla.go:276-284, :76-78), hex so that the line format stays one per name
        auto print_hex = [](const char *name, const std::string &v) {
            std::printf("%s ", name);
            for (unsigned char c : v) std::printf("%02x", c);
            std::printf("\n");
        };
        print_hex("str_est2", est2.String());
        print_hex("str_kf", kf->String());
        // a singular step: H = 0 and R = 0  =>  S = H P H' + R = 0, Inverse fails (vanilla.go:162-167)
        kf->SetMeasurementMatrix(Hzero);
        kf->SetNoise(zeroR);
        int threw = 0;
        const long long step_before_fail = (long long)kf->Step();
        try {
            (void)kf->Update(NewVector(1, {1.1}), NewVector(0));
        } catch (const StepError &e) {
            threw = (e.status & KB_ST_SINGULAR) ? 1 : 0;
            std::printf("step_error %s\n", e.what());
        }
        std::printf("singular_step_threw %d\n", threw);
        const long long step_after_fail = (long long)kf->Step();
        // the filter kept est2 (prevEst untouched), and the next valid Update succeeds without any clean-up call
        kf->SetMeasurementMatrix(H);
        kf->SetNoise(good);
        const Estimate est3 = kf->Update(NewVector(1, {1.3}), NewVector(0));
        print_vec("x3", est3.State()); print_vec("P3", est3.Covariance());
        std::printf("status3 %u\n", est3.Status()[0]);
        std::printf("vanilla_steps %lld %lld %lld\n", step_before_fail, step_after_fail, (long long)kf->Step());

        // ---- 3: SRIF 6 states / 2 measurements, singular Phi at the second step --------------------------------------
        const int n = 6, p = 2;
        Matrix P0(n, n);
        for (int i = 0; i < n; i++) P0.data[(size_t)i * n + i] = i < 3 ? 10.0 : 1.0;
        std::vector<double> x0v = {0.3, -0.1, 0.2, 0.05, -0.02, 0.01};
        SRIF srif(NewVector(n, x0v), P0, p, false, NewNoiseless(Matrix(n, n), Matrix(p, p, {1e-2, 0, 0, 1e-3})));
        auto phi = [&](double eps) {
            Matrix m = Identity(n);
            for (int i = 0; i < 3; i++) m.data[(size_t)i * n + i + 3] = 0.1;
            m.data[(size_t)4 * n + 1] = eps;
            return m;
        };
        Matrix Ht(p, n, {1, 0, 0, 0.5, 0, 0, 0, 1, 0, 0, 0.5, 0});
        srif.Prepare(phi(0.01), Ht);
        const Estimate s1 = srif.Update(NewVector(p, {0.4, -0.3}), NewVector(p, {0.35, -0.25}));
        Matrix bad = phi(0.02);
        for (int j = 0; j < n; j++) bad.data[(size_t)2 * n + j] = 0.0;   // a zero row: exactly singular
        srif.Prepare(bad, Ht);
        int sthrew = 0;
        const long long sstep_before = (long long)srif.Step();
        try {
            (void)srif.Update(NewVector(p, {0.5, -0.2}), NewVector(p, {0.45, -0.15}));
        } catch (const StepError &e) {
            sthrew = 1;
            std::printf("srif_step_error %s\n", e.what());
        }
        std::printf("srif_singular_step_threw %d\n", sthrew);
        const long long sstep_after = (long long)srif.Step();
        // the failed call returned before `kf.locked = true` (srif.go:112-114 against :158): a second Update without a new
        // Prepare() is accepted (and fails the same way), a successful one locks the filter again
        int relocked = 0, retry_threw = 0;
        try { (void)srif.Update(NewVector(p, {0.5, -0.2}), NewVector(p, {0.45, -0.15})); } catch (const StepError &) { retry_threw = 1; }
        srif.Prepare(phi(0.03), Ht);
        const Estimate s3 = srif.Update(NewVector(p, {0.6, -0.1}), NewVector(p, {0.55, -0.05}));
        try { (void)srif.Update(NewVector(p, {0.6, -0.1}), NewVector(p, {0.55, -0.05})); } catch (const Error &e) { relocked = e.code == KB_ERR_LOCKED; }
        std::printf("srif_steps %lld %lld %lld retry_threw %d relocked %d\n", sstep_before, sstep_after, (long long)srif.Step(), retry_threw, relocked);
        print_vec("srif_x1", s1.State());
        print_vec("srif_x3", s3.State()); print_vec("srif_P3", s3.Covariance());

        // ---- 5: Monte-Carlo runs and chi-square with the reference's signatures (examples/robot/main.go:31-49) -------------
        {
            const double dt = 0.1;
            Matrix Fr(2, 2, {1, dt, 0, 1}), Gr(2, 1, {0.5 * dt * dt, dt}), Hr(1, 2, {1, 0});
            Matrix Qr(2, 2, {5e-2, 5e-4, 5e-4, 1e-3}), Rr(1, 1, {0.05});
            const int sims = 6, steps = 5;
            auto mcKF = NewPurePredictorVanilla(NewVector(2, {0.7, -0.3}), ScaledIdentity(2, 2.0), Fr, Gr, Hr, NewAWGN(Qr, Rr, 4242)).first;
            auto chiKF = NewVanilla(NewVector(2, {0.0, 0.0}), ScaledIdentity(2, 2.0), Fr, Gr, Hr, NewNoiseless(Qr, Rr)).first;
            std::vector<Vector> controls;
            for (int k = 0; k < steps; k++) controls.push_back(NewVector(1, {std::cos(0.75 * (k + 1) * 0.1)}));
            const MonteCarloRuns runs = NewMonteCarloRuns(sims, steps, 1, controls, *mcKF);
            std::printf("mc_shape %lld %d %zu %zu\n", (long long)runs.runs, runs.steps, runs.Runs.size(), runs.Runs[0].Estimates.size());
            for (int r = 0; r < sims; r++)
                for (int k = 0; k < steps; k++) {
                    char name[64];
                    std::snprintf(name, sizeof(name), "mc_x_%d_%d", r, k);
                    print_vec(name, runs.Runs[(size_t)r].Estimates[(size_t)k].State());
                    std::snprintf(name, sizeof(name), "mc_y_%d_%d", r, k);
                    print_vec(name, runs.Runs[(size_t)r].Estimates[(size_t)k].Measurement());
                }
            print_vec("mc_P_3", runs.Runs[2].Estimates[3].Covariance());
            print_vec("mc_K_3", runs.Runs[2].Estimates[3].Gain());
            std::printf("mc_mean_4 %.17g %.17g\nmc_std_4 %.17g %.17g\n", runs.Mean(4)[0], runs.Mean(4)[1], runs.StdDev(4)[0], runs.StdDev(4)[1]);
            const auto csv = runs.AsCSV({"xi", "xi_dot"});
            print_hex("mc_csv_0", csv[0]);
            print_hex("mc_csv_1", csv[1]);
            const auto chi = NewChiSquare(*chiKF, runs, controls, true, true);
            std::printf("chi_nis");
            for (double v : chi.first) std::printf(" %.17g", v);
            std::printf("\nchi_nees");
            for (double v : chi.second) std::printf(" %.17g", v);
            std::printf("\n");
            std::printf("mckf_step_after %lld\n", (long long)mcKF->Step());   // left Reset() (montecarlo.go:116)
        }
    } catch (const Error &e) {
        std::fprintf(stderr, "gokalman error %d: %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}
