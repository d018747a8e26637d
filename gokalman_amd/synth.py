"""Deterministic synthetic workloads: per-filter constant-velocity
models with perturbed F/H, per-filter Q/R, a simulated truth trajectory and measurements.

Used by bench.py and the parity tests so that both run the same inputs.  Host-side numpy
only; nothing here is on the hot path.
"""
import numpy as np

SEED = 0x6B616C6D616E  # "kalman"


def linear_batch(nfilters, n=6, p=3, steps=1, seed=SEED, dtype=np.float64):
    """Config B/C inputs: returns dict(x0[N,n], P0[N,n,n], F[N,n,n], H[N,p,n], Q[N,n,n], R[N,p,p], y[T,N,p])."""
    assert n % 2 == 0 and p <= n // 2
    h = n // 2
    rng = np.random.default_rng(seed)
    N = nfilters
    dt = rng.uniform(0.05, 0.15, size=N)
    I = np.eye(h)
    F = np.zeros((N, n, n))
    F[:, :h, :h] = I
    F[:, h:, h:] = I
    F[:, :h, h:] = dt[:, None, None] * I
    F += 1e-3 * rng.standard_normal((N, n, n))
    H = np.zeros((N, p, n))
    H[:, :, :p] = np.eye(p)
    H += 1e-3 * rng.standard_normal((N, p, n))
    q = np.exp(rng.uniform(np.log(1e-6), np.log(1e-4), size=N))
    Q = np.zeros((N, n, n))
    Q[:, :h, :h] = (dt ** 3 / 3)[:, None, None] * I
    Q[:, :h, h:] = (dt ** 2 / 2)[:, None, None] * I
    Q[:, h:, :h] = (dt ** 2 / 2)[:, None, None] * I
    Q[:, h:, h:] = dt[:, None, None] * I
    Q *= q[:, None, None]
    A = rng.standard_normal((N, p, p))
    r = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    R = 1e-4 * np.einsum("nij,nkj->nik", A, A)
    R[:, np.arange(p), np.arange(p)] += r
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n))
    P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(h, 10.0), np.full(h, 1.0)])
    # truth + measurements
    LQ = np.linalg.cholesky(Q + 1e-18 * np.eye(n))
    LR = np.linalg.cholesky(R)
    xt = x0 + np.einsum("nij,nj->ni", np.linalg.cholesky(P0), rng.standard_normal((N, n)))
    y = np.zeros((steps, N, p))
    for k in range(steps):
        xt = np.einsum("nij,nj->ni", F, xt) + np.einsum("nij,nj->ni", LQ, rng.standard_normal((N, n)))
        y[k] = np.einsum("nij,nj->ni", H, xt) + np.einsum("nij,nj->ni", LR, rng.standard_normal((N, p)))
    out = dict(x0=x0, P0=P0, F=F, H=H, Q=Q, R=R, y=y)
    return {k: np.ascontiguousarray(v.astype(dtype)) for k, v in out.items()}


def rel_frobenius(a, b):
    """max over filters of ||a_i - b_i||_F / ||b_i||_F (b = reference)."""
    a = np.asarray(a, dtype=np.float64).reshape(a.shape[0], -1)
    b = np.asarray(b, dtype=np.float64).reshape(b.shape[0], -1)
    num = np.linalg.norm(a - b, axis=1)
    den = np.linalg.norm(b, axis=1)
    den = np.where(den == 0, 1.0, den)
    return float(np.max(num / den))
