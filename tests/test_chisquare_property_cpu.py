"""A pin for the oracle's NewChiSquare (chisquare.go:16-95) that does not depend on any restated formula: for a filter whose
model IS the truth's (same F, H, Q, R, Gaussian noise, Gaussian initial error with covariance P0), the normalised
innovation squared is chi-square with p degrees of freedom and the normalised estimation error squared chi-square with n
(Bar-Shalom, Li, Kirubarajan, "Estimation with Applications to Tracking and Navigation", section 5.4): their means over
the runs must be p and n at every step.  The oracle's Vanilla update is pinned by examples/jerkcar (test_oracle_golden.py)."""
import numpy as np

from oracle import oracle as orc


def test_nis_and_nees_means_are_p_and_n_for_a_consistent_filter():
    rng = np.random.default_rng(2016)
    n, p, runs, steps = 4, 2, 600, 12
    dt = 0.1
    F = np.eye(n); F[0, 2] = F[1, 3] = dt
    H = np.zeros((p, n)); H[0, 0] = H[1, 1] = 1.0
    q = 1e-2
    Q = q * np.array([[dt ** 3 / 3, 0, dt ** 2 / 2, 0], [0, dt ** 3 / 3, 0, dt ** 2 / 2], [dt ** 2 / 2, 0, dt, 0], [0, dt ** 2 / 2, 0, dt]])
    R = np.diag([4e-3, 9e-3])
    x0, P0 = np.array([1.0, -0.5, 0.3, 0.2]), np.diag([0.5, 0.5, 0.1, 0.1])
    LQ, LR, LP = np.linalg.cholesky(Q), np.linalg.cholesky(R), np.linalg.cholesky(P0)
    truth_x = np.zeros((runs, steps, n)); truth_y = np.zeros((runs, steps, p))
    for r in range(runs):
        x = x0 + LP @ rng.standard_normal(n)          # the filter starts from (x0, P0): the truth is one draw from that prior
        for t in range(steps):
            x = F @ x + LQ @ rng.standard_normal(n)
            truth_x[r, t] = x
            truth_y[r, t] = H @ x + LR @ rng.standard_normal(p)
    def make():
        f = orc.Filter.ldkf(orc.VANILLA, x0, P0, F, None, H, Q, R)
        f._H, f._R = H, R   # kf.GetMeasurementMatrix() / kf.GetNoise().MeasurementMatrix() for the NIS (chisquare.go:64-68)
        return f
    nis, nees = orc.chisquare(make, truth_x, truth_y, None)
    # standard error of a chi-square(k) mean over `runs` samples: sqrt(2 k / runs); 5 sigma bands
    assert np.all(np.abs(nis - p) < 5 * np.sqrt(2 * p / runs)), nis
    assert np.all(np.abs(nees - n) < 5 * np.sqrt(2 * n / runs)), nees
    # and over all steps together (steps x runs samples; correlated along a run, so a loose 6 sigma)
    assert abs(nis.mean() - p) < 6 * np.sqrt(2 * p / (runs * steps)) * 2
    assert abs(nees.mean() - n) < 6 * np.sqrt(2 * n / runs)
