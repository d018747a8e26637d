"""kb_van_loan (c2d.go:13-75) on the GPU against the oracle, through the C ABI."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-9     # relative Frobenius, fp64


def _systems(rng, N, n, q, scale, dtmax=0.5):
    A = rng.standard_normal((N, n, n)) * scale
    G = rng.standard_normal((N, n, q))
    L = rng.standard_normal((N, q, q))
    W = L @ np.swapaxes(L, 1, 2) + 0.1 * np.eye(q)
    dt = rng.uniform(0.01, dtmax, N)
    return A, G, W, dt


def test_reference_test_vectors():
    # c2d_test.go:10-32
    F, Q, st = ga.van_loan([[0, 1], [0, 0]], [[0], [1]], [[1]], 0.1)
    assert st == 0
    assert np.allclose(F, [[1, 0.1], [0, 1]], atol=1e-3, rtol=1e-3)
    assert np.allclose(Q, [[0.0003, 0.005], [0.005, 0.1]], atol=1e-3, rtol=1e-3)
    assert np.allclose(Q, [[1e-3 / 3, 5e-3], [5e-3, 0.1]], rtol=1e-13)
    _, _, st = ga.van_loan([[1, 1], [0, 1]], [[0], [1]], [[1]], 10)
    assert st & k.ST_NYQUIST


@pytest.mark.parametrize("n,q", [(1, 1), (2, 1), (3, 2), (4, 2), (6, 3), (8, 4)])
@pytest.mark.parametrize("scale", [0.3, 20.0])
def test_batch_matches_oracle(n, q, scale):
    rng = np.random.default_rng(100 * n + q)
    N = 257                                           # ragged last wave
    A, G, W, dt = _systems(rng, N, n, q, scale, dtmax=0.5 if scale < 1 else 0.06)   # scale 20: |M|_1 ~ 5..20, 1-2 squarings
    F, Q, st = ga.van_loan(A, G, W, dt)
    worstF = worstQ = 0.0
    for i in range(N):
        rc, Fo, Qo = orc.van_loan(A[i], G[i], W[i], dt[i])
        worstF = max(worstF, np.linalg.norm(F[i] - Fo) / np.linalg.norm(Fo))
        # Q = F * (F^-1 Q) (c2d.go:71): for |A dt| >> 1 the two factors are huge and the product cancels; both sides
        # match factor by factor, so the bar for Q scales with that cancellation, kappa = |F| |F^-1 Q| / |Q| (1 for scale 0.3)
        kappa = max(1.0, np.linalg.norm(Fo) * np.linalg.norm(np.linalg.solve(Fo, Qo)) / np.linalg.norm(Qo))
        worstQ = max(worstQ, np.linalg.norm(Q[i] - Qo) / np.linalg.norm(Qo) / kappa)
        assert bool(st[i] & k.ST_NYQUIST) == bool(rc & 1), (i, st[i], rc)
        assert bool(st[i] & k.ST_ASYMMETRIC) == bool(rc & 2), (i, st[i], rc)
        assert np.array_equal(Q[i], Q[i].T)
    assert worstF <= TOL and worstQ <= TOL, (worstF, worstQ)


def test_broadcast_model_and_fp32():
    rng = np.random.default_rng(9)
    A, G, W, dt = _systems(rng, 1, 4, 2, 0.5)
    dts = np.linspace(0.01, 1.0, 100)
    F, Q, st = ga.van_loan(A[0], G[0], W[0], dts)
    F32, Q32, _ = ga.van_loan(A[0], G[0], W[0], dts, dtype=k.F32)
    for i in (0, 50, 99):
        _, Fo, Qo = orc.van_loan(A[0], G[0], W[0], dts[i])
        assert np.linalg.norm(F[i] - Fo) <= TOL * np.linalg.norm(Fo)
        assert np.linalg.norm(Q[i] - Qo) <= TOL * np.linalg.norm(Qo)
        assert np.linalg.norm(F32[i] - Fo) <= 1e-5 * np.linalg.norm(Fo)
        assert np.linalg.norm(Q32[i] - Qo) <= 1e-4 * np.linalg.norm(Qo)


def test_device_variant_feeds_a_filter_batch():
    """kb_van_loan_dev writes F, Q in the planar layout kb_set_dev reads: continuous models -> filter, no host hop."""
    import torch
    rng = np.random.default_rng(11)
    N, n, q, p = 200, 4, 2, 2
    A, G, W, dt = _systems(rng, N, n, q, 0.3)
    dev = torch.device("cuda:0")
    planar = lambda a: torch.tensor(np.ascontiguousarray(a.reshape(N, -1).T), device=dev)
    dA, dG, dW, ddt = planar(A), planar(G), planar(W), torch.tensor(dt, device=dev)
    dF, dQ = torch.zeros((n * n, N), dtype=torch.float64, device=dev), torch.zeros((n * n, N), dtype=torch.float64, device=dev)
    dst = torch.zeros(N, dtype=torch.int32, device=dev)
    H = np.hstack([np.eye(p), np.zeros((p, n - p))])
    R = 0.01 * np.eye(p)
    b = ga.FilterBatch(k.VANILLA, n, p, 0, N)
    b.set(k.X, np.ones(n), 1); b.set(k.P, np.eye(n), 2); b.set(k.H, H, 2, p_rows=p); b.set(k.R, R, 2, p_rows=p)
    torch.cuda.synchronize()
    k.check(k.lib().kb_van_loan_dev(k.F64, n, q, N, dA.data_ptr(), dG.data_ptr(), dW.data_ptr(), ddt.data_ptr(), N,
                                    dF.data_ptr(), dQ.data_ptr(), dst.data_ptr(), b.stream()))
    b.set_dev(k.F, dF.data_ptr(), N)
    b.set_dev(k.Q, dQ.data_ptr(), N)
    b.init()
    y = rng.standard_normal((N, p))
    est = b.update(y)
    x, P = est.state(), est.covariance()
    for i in (0, 77, 199):
        _, Fo, Qo = orc.van_loan(A[i], G[i], W[i], dt[i])
        f = orc.Filter.ldkf(orc.VANILLA, np.ones(n), np.eye(n), Fo, None, H, Qo, R)
        assert f.update(y[i]) == orc.OK
        assert np.linalg.norm(x[i] - f.state()) <= TOL * np.linalg.norm(f.state())
        assert np.linalg.norm(P[i] - f.covariance()) <= TOL * np.linalg.norm(f.covariance())
