"""An independent pin for the oracle's SRIF time update + whitening (srif.go:111-148), which no reference fixture
exercises without the external `smd` propagator: on a quirk-neutral case the SRIF recursion must reproduce the
Information filter's, and the oracle's Information filter IS pinned (examples/jerkcar/information.csv, tests/
test_oracle_golden.py).

Quirk-neutral: measurement noise R = identity, so chol_L(R) = L = I and it does not matter that srif.go:48 keeps L where
L^-1 was meant; no process noise (the Information oracle gets Q = 1e-30 I, i.e. Q^-1 = 1e30 I, which makes its
Z = -M (M + Q^-1)^-1 vanish to ~1e-30 relative).  Then, with y = real - computed as the Information filter's measurement,
    R_k^T R_k == I_k          (information matrix)
    R_k^T b_k == i_k          (information vector)
    R_k^-1 b_k == I_k^-1 i_k  (state)
for every step, Predict() included -- for a general and for a triangular Phi."""
import numpy as np
import pytest

from oracle import oracle as orc


def _run(n, p, steps, seed, triangular):
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal(n)
    P0 = np.diag(np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)]))
    Rn = np.eye(p)
    srif = orc.Filter.srif(x0, P0, Rn, p)
    info = None
    out = []
    for t in range(steps):
        Phi = np.eye(n) + 5e-2 * rng.standard_normal((n, n))
        if triangular:
            Phi = np.triu(Phi)
        Ht = rng.standard_normal((p, n))
        real = rng.standard_normal(p)
        comp = real + 1e-1 * rng.standard_normal(p)
        predict = t == 2
        if info is None:
            info = orc.Filter.information_from_state(x0, P0, Phi, None, Ht, 1e-30 * np.eye(n), Rn)
        info.set_state_transition(Phi)     # refreshes F^-1 (information.go:117-123)
        srif.prepare(Phi, Ht)
        if predict:
            # Predict(): time update only.  The Information filter has no Predict(); a measurement with H = 0 adds nothing
            # (I+ = I- + H^T R^-1 H, i+ = i- + H^T R^-1 y)
            info.set_measurement_matrix(np.zeros((p, n)))
            assert srif.predict_nl() == orc.OK
            assert info.update(np.zeros(p)) == orc.OK
        else:
            info.set_measurement_matrix(Ht)
            assert srif.update_nl(real, comp) == orc.OK
            assert info.update(real - comp) == orc.OK
        out.append((srif.raw_mat(), srif.raw_vec(), srif.state(), info.raw_mat(), info.raw_vec(), info.state()))
    return out


@pytest.mark.parametrize("triangular", [False, True])
@pytest.mark.parametrize("n,p", [(6, 2), (12, 6), (4, 1)])
def test_srif_recursion_equals_the_pinned_information_filter(n, p, triangular):
    for seed in range(5):
        for t, (R, b, xs, I, i, xi) in enumerate(_run(n, p, 6, 100 * n + seed, triangular)):
            scale = np.linalg.norm(I)
            assert np.linalg.norm(R.T @ R - I) <= 1e-10 * scale, (seed, t)
            assert np.linalg.norm(R.T @ b - i) <= 1e-10 * max(np.linalg.norm(i), 1.0), (seed, t)
            assert np.linalg.norm(xs - xi) <= 1e-9 * max(np.linalg.norm(xi), 1.0), (seed, t)


def test_srif_whitening_uses_chol_l_of_r_not_its_inverse():
    """The quirk itself (srif.go:48), against the same pinned Information algebra: with R = diag(r) the SRIF adds
    H^T (L^T L) H = H^T R H to the information matrix where a correct filter adds H^T R^-1 H."""
    n, p = 6, 2
    rng = np.random.default_rng(3)
    x0 = rng.standard_normal(n)
    P0 = np.diag([10.0, 10, 10, 1, 1, 1])
    r = np.array([4.0, 0.25])
    srif = orc.Filter.srif(x0, P0, np.diag(r), p)
    Phi = np.eye(n) + 5e-2 * rng.standard_normal((n, n))
    Ht = rng.standard_normal((p, n))
    srif.prepare(Phi, Ht)
    assert srif.update_nl(rng.standard_normal(p), rng.standard_normal(p)) == orc.OK
    R1 = srif.raw_mat()
    Phi_inv = np.linalg.inv(Phi)
    I_bar = Phi_inv.T @ np.diag(1.0 / np.diag(P0)) @ Phi_inv
    quirk = I_bar + Ht.T @ np.diag(r) @ Ht
    correct = I_bar + Ht.T @ np.diag(1.0 / r) @ Ht
    assert np.linalg.norm(R1.T @ R1 - quirk) <= 1e-10 * np.linalg.norm(quirk)
    assert np.linalg.norm(R1.T @ R1 - correct) >= 1e-2 * np.linalg.norm(correct)
