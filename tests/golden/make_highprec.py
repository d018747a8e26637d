"""Generates tests/golden/generated/*_hp.npz: an ARBITER ABOVE fp64 for the ill-conditioned parity legs (task 4).

Build container only (mpmath; 60 significant digits).  Each function below restates one Update of the reference, statement by
statement, in multi-precision arithmetic -- inputs rounded to fp64 FIRST (they are the very arrays the engine and the oracle get),
outputs rounded to fp64 LAST -- so that for problems on which two correct fp64 evaluations drift apart by 1e-5 (R = 1e-6 against
P0 = 10) there is an answer to "which one is closer to the exact result":

  hybrid_update      hybrid.go:104-204      (CKF and EKF; Prepare()d Phi / Htilde per step, no SNC)
  vanilla_update     vanilla.go:128-220     with BatchNoise (noise.go:67-106: recorded vectors, ZERO Q and R matrices)
  srif_update        srif.go:101-160, :298-340, helper.go:142-172  (time update + whitening + HouseholderTransf: a second, independent
                                             reading beside the oracle's -- no reference fixture reaches the time update)
  squareroot_update  squareroot.go:129-274  (QR with LAPACK's dlarfg sign convention: with the reference's untransposed Uc the
                                             result DEPENDS on the signs of R's diagonal, so the convention is part of the algorithm)

The reference is Go and cannot run here: these are not reference outputs.  What pins the restatements is (a) the statement-by-statement
reading cited per line and (b) tests/test_highprec_cpu.py: on WELL-conditioned inputs the oracle (itself pinned to the reference's
jerkcar CSVs) agrees with them to ~1e-14.  Run from the repo root:   python tests/golden/make_highprec.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mpmath as mp  # noqa: E402

from gokalman_amd import synth  # noqa: E402

mp.mp.dps = 60
OUT = os.path.join(ROOT, "tests", "golden", "generated")
SINGULAR_BELOW = mp.mpf(10) ** -40   # |pivot| / scale: exactly singular in exact arithmetic (working precision 1e-60)


def M(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return mp.matrix([[mp.mpf(float(v)) for v in row] for row in a])


def to_np(m):
    return np.array([[float(m[i, j]) for j in range(m.cols)] for i in range(m.rows)], dtype=np.float64)


def as_sym(m):
    """helper.go:65-84 AsSymDense keeps every entry of the Dense, NewSymDense then reads the UPPER triangle."""
    s = m.copy()
    for i in range(m.rows):
        for j in range(i):
            s[i, j] = m[j, i]
    return s


def inverse_or_none(S, scale):
    """mat64 Dense.Inverse: LU with partial pivoting; an exactly zero pivot is the error return.  In 60-digit arithmetic 'exactly zero'
    is a pivot below 1e-40 of the problem's scale."""
    n = S.rows
    A = S.copy()
    for j in range(n):
        piv = max(range(j, n), key=lambda r: abs(A[r, j]))
        if abs(A[piv, j]) <= SINGULAR_BELOW * scale:
            return None
        if piv != j:
            for c in range(n):
                A[j, c], A[piv, c] = A[piv, c], A[j, c]
        for r in range(j + 1, n):
            f = A[r, j] / A[j, j]
            for c in range(j, n):
                A[r, c] -= f * A[j, c]
    return mp.inverse(S)


def qr_r(A):
    """R of mat64.QR.Factorize = LAPACK dgeqr2 / dlarfg: beta = -sign(alpha) |(alpha, x)|, and H = I when x == 0 (the diagonal entry
    keeps its sign).  Returns the m x n upper-trapezoidal R."""
    A = A.copy()
    m, n = A.rows, A.cols
    for j in range(min(m, n)):
        alpha = A[j, j]
        xn2 = mp.fsum(A[i, j] ** 2 for i in range(j + 1, m))
        if xn2 == 0:
            continue
        beta = -mp.sign(alpha) * mp.sqrt(alpha * alpha + xn2) if alpha != 0 else -mp.sqrt(xn2)
        tau = (beta - alpha) / beta
        v = [mp.mpf(1)] + [A[i, j] / (alpha - beta) for i in range(j + 1, m)]
        for c in range(j + 1, n):
            w = mp.fsum(v[i - j] * A[i, c] for i in range(j, m))
            for i in range(j, m):
                A[i, c] -= tau * v[i - j] * w
        A[j, j] = beta
        for i in range(j + 1, m):
            A[i, j] = mp.mpf(0)
    return A


def hybrid_update(x, P, Phi, Ht, R, real, comp, ekf):
    """hybrid.go:104-204 without SNC.  Returns (x, P) or None when H P- H^T + R is exactly singular."""
    PBar = Phi * P * Phi.T                                   # :116-118
    PHt = PBar * Ht.T                                        # :147
    S = Ht * PHt + R                                         # :148-149
    Sinv = inverse_or_none(S, max(abs(v) for v in S) if S.rows else mp.mpf(1))
    if Sinv is None:
        return None                                          # :150-152
    K = PHt * Sinv                                           # :153
    y = real - comp                                          # :156-157
    if ekf:
        xHat = K * y                                         # :160-161
    else:
        xBar = Phi * x                                       # :164-165
        innov = y - Ht * xBar                                # :167-169
        xHat = xBar + K * innov                              # :171-172
    n = P.rows
    IKH = mp.eye(n) - K * Ht                                 # :175-177
    Pn = IKH * PBar * IKH.T + K * R * K.T                    # :178-182
    return xHat, as_sym(Pn)                                  # :189-192


def vanilla_update(x, P, F, H, Q, R, y, w_proc, v_meas):
    """vanilla.go:128-220, no control; Noise.Process(k) = w_proc is added TWICE (:146 and :195, the reference's behaviour),
    Noise.Measurement(k) only enters yhat.  Returns (x, P) or None (singular S)."""
    xm = F * x + w_proc                                      # :139-146
    Pm = F * P * F.T + Q                                     # :149-152
    PHt = Pm * H.T                                           # :161
    S = H * PHt + R                                          # :162-163
    scale = max([abs(v) for v in H * (F * F.T) * H.T] + [mp.mpf(0)])   # what S would be for P = I: the problem's own scale
    Sinv = inverse_or_none(S, scale)
    if Sinv is None:
        return None                                          # :164-167
    K = PHt * Sinv                                           # :168
    innov = y - H * xm                                       # :183-184
    xp = xm + K * innov + w_proc                             # :185-195
    n = P.rows
    IKH = mp.eye(n) - K * H                                  # :198-200
    Pp = IKH * Pm * IKH.T + K * R * K.T                      # :201-205
    return xp, as_sym(Pp)                                    # :212-215


def squareroot_init(P0, Q, R):
    """squareroot.go:36-42 (stddev = chol_L(P0)) and SetNoise :102-113 (sqrtQ, sqrtR = lower Cholesky factors)."""
    return mp.cholesky(P0), mp.cholesky(Q), mp.cholesky(R)


def squareroot_update(x, S, F, H, sqrtQ, sqrtR, y):
    """squareroot.go:129-274, Noiseless, no control.  Returns (x, stddev); Covariance() = stddev stddev^T (:316-325)."""
    n, p = x.rows, y.rows
    xm = F * x                                               # :141-148
    C = mp.matrix(2 * n, n)
    sTFT = S.T * F.T                                         # :160
    for i in range(n):
        for j in range(n):
            C[i, j] = sTFT[i, j]                             # :163-168
            C[n + i, j] = sqrtQ.T[i, j]                      # :170-176
    Uc = qr_r(C)                                             # :177-181
    Sm = Uc[0:n, 0:n]                                        # :187  (QUIRK: the upper-triangular factor itself, not its transpose)
    SmTHT = Sm.T * H.T                                       # :192-193
    D = mp.matrix(n + p, n + p)                              # :197-218, by columns
    for c in range(n + p):
        for r in range(n + p):
            if c < p:
                D[r, c] = sqrtR.T[r, c] if r < p else SmTHT[r - p, c]
            elif r < p:
                D[r, c] = mp.mpf(0)
            else:
                D[r, c] = Sm.T[r - p, c - p]
    UD = qr_r(D)                                             # :221-224
    SplusT = UD[p:n + p, p:n + p]                            # :229
    SyyT = UD[0:p, 0:p]                                      # :230
    WT = UD[0:p, p:n + p]                                    # :231
    Syy, W = SyyT.T, WT.T                                    # :233-236
    K = W * mp.inverse(Syy)                                  # :244-254 (the error of Inverse is never looked at: `err`, not `invErr`)
    innov = y - H * xm                                       # :257-259
    xp = xm + K * innov                                      # :260-270 (Noiseless: Process(k) = 0)
    return xp, SplusT.T


def srif_init(x0, P0, R):
    """srif.go:14-49: I0 = diag(1 / P0_ii), R0 = chol_L(I0), b0 = R0 x0; QUIRK :47: the filter keeps chol_L(R) (sqrtMeasNoise) in the field
    named sqrtInvNoise -- the inverse formed at :41-44 is discarded."""
    n = P0.rows
    I0 = mp.matrix(n, n)
    for i in range(n):
        I0[i, i] = 1 / P0[i, i]
    R0 = mp.cholesky(I0)
    return R0 * x0, R0, mp.cholesky(R)


def householder_transf(A, n, m):
    """helper.go:142-172, Sign() of :133-138 with its 1e-12 dead band."""
    for k in range(n):
        sigma = mp.sqrt(mp.fsum(A[i, k] ** 2 for i in range(k, m + n)))
        akk = A[k, k]
        sigma *= mp.mpf(1) if abs(akk) <= mp.mpf("1e-12") else mp.sign(akk)
        u = [mp.mpf(0)] * (m + n)
        u[k] = akk + sigma
        A[k, k] = -sigma
        for i in range(k + 1, m + n):
            u[i] = A[i, k]
        beta = 1 / (sigma * u[k])
        for j in range(k + 1, n + 1):
            gamma = mp.fsum(u[i] * A[i, j] for i in range(k, m + n)) * beta
            for i in range(k, m + n):
                A[i, j] -= gamma * u[i]
            for i in range(k + 1, m + n):
                A[i, k] = mp.mpf(0)


def srif_update(b, R, Phi, Ht, L, real, comp):
    """srif.go:101-160 (Update) with measurementSRIFUpdate :298-340.  Returns (b_k, R_k)."""
    n, m = b.rows, real.rows
    invPhi = mp.inverse(Phi)                                 # :111-114
    RBar = R * invPhi                                        # :115
    xBar = Phi * (mp.inverse(R) * b)                         # :117-118, State() :223-234
    bBar = RBar * xBar                                       # :119
    y = L * (real - comp)                                    # :143-148
    Hw = L * Ht
    A = mp.matrix(m + n, n + 1)                              # :309-321
    for i in range(n):
        for j in range(n):
            A[i, j] = RBar[i, j]
        A[i, n] = bBar[i]
    for i in range(m):
        for j in range(n):
            A[n + i, j] = Hw[i, j]
        A[n + i, n] = y[i]
    householder_transf(A, n, m)                              # :323
    return A[0:n, n:n + 1], A[0:n, 0:n]                      # :326-331


def gen_srif(name, n, p, N=32, T=10, seed=77):
    """config E's problem (bench.py's generator: Phi = I + 1e-2 randn, Htilde = randn, R diagonal 1e-4 .. 1e-2, P0 = diag(10.., 1..)) in
    fp64 inputs: a SECOND, independent reading of srif.go for the oracle's SRIF (whose time update no reference fixture reaches:
    srif_test.go needs the external `smd` package) and the exact answer the fp32 kernel's achieved error is quoted against."""
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    Phi = np.eye(n) + 1e-2 * rng.standard_normal((T, N, n, n))
    Ht = rng.standard_normal((T, N, p, n))
    real = rng.standard_normal((T, N, p)); comp = real + 1e-2 * rng.standard_normal((T, N, p))
    # (the fp32 kernels get these very values: rounded to fp32 first, so that fp32 and fp64 runs see the same problem)
    Phi, Ht, real, comp = [a.astype(np.float32).astype(np.float64) for a in (Phi, Ht, real, comp)]
    bs, Rs = np.zeros((T, N, n)), np.zeros((T, N, n, n))
    for i in range(N):
        b, Rm, L = srif_init(M(x0[i]), M(P0[i]), M(R[i]))
        for t in range(T):
            b, Rm = srif_update(b, Rm, M(Phi[t, i]), M(Ht[t, i]), L, M(real[t, i]), M(comp[t, i]))
            bs[t, i], Rs[t, i] = to_np(b)[:, 0], to_np(Rm)
    np.savez_compressed(os.path.join(OUT, name + "_hp.npz"), x0=x0, P0=P0, R=R, Phi=Phi, Ht=Ht, real=real, comp=comp, b=bs, Rk=Rs,
                        digits=np.array(mp.mp.dps))
    print(name, "done")


def rows(a):
    return [M(v) for v in a]


def gen_hybrid(name, phi_kind, ekf, N=64, T=20, seed=2016):
    """configs[3] D(ii) (n = 6, p = 2, R = diag(1e-6, 1e-6), P0 = diag(10, 10, 10, 1, 1, 1), hybrid_test.go:174-180).
    phi_kind "bench": Phi_k = I + 1e-2 randn, Htilde = randn (what bench.py's leg times); "stm": SURVEY's two-body-like
    Phi_k = [[I, dt I], [-w^2 dt I, I]] with random unit-row Htilde."""
    n, p = 6, 2
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10, 10, 10, 1, 1, 1]
    R = np.diag([1e-6, 1e-6])
    if phi_kind == "bench":
        Phi = np.eye(n) + 1e-2 * rng.standard_normal((T, N, n, n))
        Ht = rng.standard_normal((T, N, p, n))
    else:
        dt = rng.uniform(5.0, 15.0, size=(T, N)); w2 = rng.uniform(1e-7, 2e-6, size=(T, N))   # ~LEO mean motion squared, 10 s steps
        Phi = np.zeros((T, N, n, n))
        I3 = np.eye(3)
        Phi[:, :, :3, :3] = I3; Phi[:, :, 3:, 3:] = I3
        Phi[:, :, :3, 3:] = dt[..., None, None] * I3
        Phi[:, :, 3:, :3] = -(w2 * dt)[..., None, None] * I3
        Ht = rng.standard_normal((T, N, p, n))
        Ht /= np.linalg.norm(Ht, axis=-1, keepdims=True)
    real = rng.standard_normal((T, N, p))
    comp = real + 1e-3 * rng.standard_normal((T, N, p))
    xs, Ps = np.zeros((T, N, n)), np.zeros((T, N, n, n))
    Rm = M(R)
    for i in range(N):
        x, P = M(x0[i]), M(P0[i])
        for t in range(T):
            out = hybrid_update(x, P, M(Phi[t, i]), M(Ht[t, i]), Rm, M(real[t, i]), M(comp[t, i]), ekf)
            assert out is not None, (name, i, t)
            x, P = out
            xs[t, i], Ps[t, i] = to_np(x)[:, 0], to_np(P)
    np.savez_compressed(os.path.join(OUT, name + "_hp.npz"), x0=x0, P0=P0, R=R, Phi=Phi, Ht=Ht, real=real, comp=comp, x=xs, P=Ps,
                        ekf=np.array(ekf), digits=np.array(mp.mp.dps))
    print(name, "done")


def gen_batchnoise(name, n, p, N=64, T=8, seed=900):
    """Vanilla + BatchNoise past n measurements: Q = R = 0 (noise.go:89-98), so after ceil(n / p) steps the exact covariance IS the zero
    matrix and from the next step on H P- H^T + R is exactly singular: the reference's Update returns an error there in exact
    arithmetic.  `valid[t, i]`: the exact step exists; x, P are recorded for those steps only (NaN elsewhere)."""
    d = synth.linear_batch(N, n, p, T, seed=seed + n)
    rng = np.random.default_rng(n)
    proc, meas = 1e-2 * rng.standard_normal((T, n)), 1e-2 * rng.standard_normal((T, p))
    Z_n, Z_p = mp.matrix(n, n), mp.matrix(p, p)
    xs, Ps = np.full((T, N, n), np.nan), np.full((T, N, n, n), np.nan)
    valid = np.zeros((T, N), dtype=bool)
    for i in range(N):
        x, P = M(d["x0"][i]), M(d["P0"][i])
        F, H = M(d["F"][i]), M(d["H"][i])
        for t in range(T):
            out = vanilla_update(x, P, F, H, Z_n, Z_p, M(d["y"][t, i]), M(proc[t]), M(meas[t]))
            if out is None:
                break
            x, P = out
            xs[t, i], Ps[t, i], valid[t, i] = to_np(x)[:, 0], to_np(P), True
    np.savez_compressed(os.path.join(OUT, name + "_hp.npz"), x0=d["x0"], P0=d["P0"], F=d["F"], H=d["H"], y=d["y"], proc=proc, meas=meas,
                        x=xs, P=Ps, valid=valid, digits=np.array(mp.mp.dps))
    print(name, "done; exact steps per filter:", sorted(set(valid.sum(axis=0).tolist())))


def gen_illcond_ldkf(name, N=64, T=20, seed=4321, well=False):
    """The headline shape (6 / 3, per-filter models of synth.linear_batch) with R = 1e-6 I against P0 = diag(10, 10, 10, 1, 1, 1): the
    linear twin of D(ii), for Vanilla (vanilla.go:128-220, Noiseless) and SquareRoot (squareroot.go:129-274).  well=True keeps the
    generator's own R (1e-4 .. 1e-2): the WELL-conditioned control that pins these restatements to the oracle."""
    n, p = 6, 3
    d = synth.linear_batch(N, n, p, T, seed=seed)
    if not well:
        d["R"] = np.ascontiguousarray(np.broadcast_to(1e-6 * np.eye(p), (N, p, p)))
    xv, Pv = np.zeros((T, N, n)), np.zeros((T, N, n, n))
    xq, Pq = np.zeros((T, N, n)), np.zeros((T, N, n, n))
    zn, zp = mp.matrix(n, 1), mp.matrix(p, 1)
    for i in range(N):
        F, H, Q, R = M(d["F"][i]), M(d["H"][i]), M(d["Q"][i]), M(d["R"][i])
        x, P = M(d["x0"][i]), M(d["P0"][i])
        S, sQ, sR = squareroot_init(P, Q, R)
        xs = x.copy()
        for t in range(T):
            y = M(d["y"][t, i])
            x, P = vanilla_update(x, P, F, H, Q, R, y, zn, zp)
            xs, S = squareroot_update(xs, S, F, H, sQ, sR, y)
            xv[t, i], Pv[t, i] = to_np(x)[:, 0], to_np(P)
            xq[t, i], Pq[t, i] = to_np(xs)[:, 0], to_np(as_sym(S * S.T))
    np.savez_compressed(os.path.join(OUT, name + "_hp.npz"), x_vanilla=xv, P_vanilla=Pv, x_squareroot=xq, P_squareroot=Pq,
                        digits=np.array(mp.mp.dps), **d)
    print(name, "done")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    jobs = {
        "ldkf_wellcond_6x3": lambda: gen_illcond_ldkf("ldkf_wellcond_6x3", N=16, T=20, well=True),
        "ldkf_illcond_6x3": lambda: gen_illcond_ldkf("ldkf_illcond_6x3"),
        "hybrid_ekf_bench_6x2": lambda: gen_hybrid("hybrid_ekf_bench_6x2", "bench", True),
        "hybrid_ckf_bench_6x2": lambda: gen_hybrid("hybrid_ckf_bench_6x2", "bench", False),
        "hybrid_ekf_stm_6x2": lambda: gen_hybrid("hybrid_ekf_stm_6x2", "stm", True),
        "srif_12x6": lambda: gen_srif("srif_12x6", 12, 6),
        "srif_7x3": lambda: gen_srif("srif_7x3", 7, 3, N=16, T=8),
        "vanilla_batchnoise_6x3": lambda: gen_batchnoise("vanilla_batchnoise_6x3", 6, 3),
        "vanilla_batchnoise_12x6": lambda: gen_batchnoise("vanilla_batchnoise_12x6", 12, 6),
    }
    for name, job in jobs.items():
        if not only or name in only:
            job()
