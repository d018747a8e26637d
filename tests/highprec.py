"""Loader + error metrics for the multi-precision fixtures tests/golden/generated/*_hp.npz (made by tests/golden/make_highprec.py):
the arbiter above fp64 for the ill-conditioned parity legs.  The rule (task 4): an engine entry point passes when its
error against the EXACT result is at most 4 x the error of the oracle (the reference-order fp64 evaluation) against the same exact result,
worst filter and median filter alike -- with a floor of a few ulps for the cases in which the oracle happens to be exact."""
import os

import numpy as np

GEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generated")
FACTOR = 4.0
FLOOR = 1e-13


def load(name):
    return np.load(os.path.join(GEN, name + "_hp.npz"))


def rel_err(got, exact, scale=None):
    """Per-filter relative Frobenius error [N]; `scale` (per filter or scalar) replaces |exact| where the exact value is (near) zero."""
    got, exact = np.asarray(got, dtype=np.float64), np.asarray(exact, dtype=np.float64)
    N = exact.shape[0]
    num = np.linalg.norm((got - exact).reshape(N, -1), axis=1)
    den = np.linalg.norm(exact.reshape(N, -1), axis=1) if scale is None else np.broadcast_to(np.asarray(scale, dtype=np.float64), (N,))
    return num / den


def summary(err):
    err = np.asarray(err)
    return {"max": float(err.max()), "median": float(np.median(err))}


def passes(engine, oracle):
    """engine / oracle: per-filter error arrays against the exact result."""
    e, o = summary(engine), summary(oracle)
    return e["max"] <= FACTOR * o["max"] + FLOOR and e["median"] <= FACTOR * o["median"] + FLOOR


def oracle_hybrid(orc, z, upto=None):
    """The oracle on a hybrid fixture: (x[T, N, n], P[T, N, n, n])."""
    T, N = z["Phi"].shape[:2]
    T = upto or T
    n, p = z["x0"].shape[1], z["real"].shape[2]
    xs, Ps = np.zeros((T, N, n)), np.zeros((T, N, n, n))
    for i in range(N):
        f = orc.Filter.hybrid(z["x0"][i], z["P0"][i], None, z["R"], p)
        if bool(z["ekf"]):
            f.enable_ekf()
        for t in range(T):
            f.prepare(z["Phi"][t, i], z["Ht"][t, i])
            assert f.update_nl(z["real"][t, i], z["comp"][t, i]) == orc.OK
            xs[t, i], Ps[t, i] = f.state(), f.covariance()
    return xs, Ps


def oracle_ldkf(orc, kind, z, upto=None):
    T, N = z["y"].shape[:2]
    T = upto or T
    n = z["x0"].shape[1]
    xs, Ps = np.zeros((T, N, n)), np.zeros((T, N, n, n))
    for i in range(N):
        f = orc.Filter.ldkf(kind, z["x0"][i], z["P0"][i], z["F"][i], None, z["H"][i], z["Q"][i], z["R"][i])
        for t in range(T):
            assert f.update(z["y"][t, i]) == orc.OK
            xs[t, i], Ps[t, i] = f.state(), f.covariance()
    return xs, Ps


def oracle_batchnoise(orc, z):
    """Vanilla + BatchNoise on the oracle: (x[T, N, n], P[T, N, n, n], rc[T, N])."""
    T, N = z["y"].shape[:2]
    n, p = z["x0"].shape[1], z["y"].shape[2]
    ZQ, ZR = np.zeros((n, n)), np.zeros((p, p))
    xs, Ps, rcs = np.zeros((T, N, n)), np.zeros((T, N, n, n)), np.zeros((T, N), dtype=np.int64)
    for i in range(N):
        f = orc.Filter.ldkf(orc.VANILLA, z["x0"][i], z["P0"][i], z["F"][i], None, z["H"][i], ZQ, ZR)
        for t in range(T):
            rcs[t, i] = f.update(z["y"][t, i], None, z["proc"][t], z["meas"][t], z["proc"][t])
            xs[t, i], Ps[t, i] = f.state(), f.covariance()
    return xs, Ps, rcs
