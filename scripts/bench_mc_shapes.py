import sys, time
import numpy as np
sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k
for (n, p, runs, steps) in [(12, 6, 1 << 18, 200), (16, 8, 1 << 18, 200), (8, 4, 1 << 18, 200), (6, 3, 1 << 18, 200)]:
    rng = np.random.default_rng(n)
    F = np.eye(n) + 0.01 * rng.standard_normal((n, n)); H = rng.standard_normal((p, n))
    A = 0.1 * rng.standard_normal((n, n)); Q = A @ A.T + 1e-3 * np.eye(n); R = 1e-2 * np.eye(p)
    truth = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, np.ones(n), np.eye(n), F, None, H, Q, R, nfilters=runs, noise=k.NOISE_AWGN, seed=3)
    mc = ga.new_monte_carlo_runs(runs, steps, p, np.zeros((1, 1)), truth, keep_runs=False)
    t0 = time.perf_counter()
    mc = ga.new_monte_carlo_runs(runs, steps, p, np.zeros((1, 1)), truth, keep_runs=False)
    dt = time.perf_counter() - t0
    print("MC n=%d: %d runs x %d steps in %.1f ms = %.2f G run-steps/s" % (n, runs, steps, dt * 1e3, runs * steps / dt / 1e9), flush=True)
for (n, p, runs, steps) in [(12, 6, 1 << 18, 200), (8, 4, 1 << 18, 200), (16, 8, 1 << 18, 200), (9, 3, 1 << 18, 200), (6, 3, 1 << 18, 200), (4, 2, 1 << 18, 200), (12, 6, 1 << 14, 50)]:
    rng = np.random.default_rng(n)
    F = np.eye(n) + 0.01 * rng.standard_normal((n, n)); H = rng.standard_normal((p, n))
    A = 0.1 * rng.standard_normal((n, n)); Q = A @ A.T + 1e-3 * np.eye(n); R = 1e-2 * np.eye(p)
    truth = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, np.ones(n), 1e-6 * np.eye(n), F, None, H, Q, R, nfilters=runs, noise=k.NOISE_AWGN, seed=3)
    kf = ga.FilterBatch.new_ldkf(k.VANILLA, np.ones(n), 1e-6 * np.eye(n), F, None, H, Q, R, nfilters=runs)
    nis, nees = ga.new_chi_square(kf, truth, np.zeros((1, 1)), steps=steps)
    t0 = time.perf_counter()
    nis, nees = ga.new_chi_square(kf, truth, np.zeros((1, 1)), steps=steps)
    dt = time.perf_counter() - t0
    print("chi-square n=%d p=%d: %d runs x %d steps in %.1f ms = %.3f G run-steps/s; mean NIS %.3f (p = %d), mean NEES %.3f (n = %d)" % (n, p, runs, steps, dt * 1e3, runs * steps / dt / 1e9, nis[5:].mean(), p, nees[5:].mean(), n), flush=True)
