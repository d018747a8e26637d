"""Generates tests/golden/generated/*.npz: small input/output vectors from the CPU oracle for the
configs of SURVEY.md section 8d (B Vanilla, C SquareRoot, Information, D Hybrid CKF/EKF, E SRIF in fp64).

The reference is Go and cannot be imported or run here, so these are NOT reference outputs: they
freeze the oracle (itself pinned to the reference's jerkcar CSVs and KATs) so that drift in the
oracle or in the HIP path shows up against committed data.  Run from the repo root:
    python tests/golden/make_generated.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gokalman_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "generated")
N, STEPS = 64, 50


def ldkf(kind, name, n, p):
    d = synth.linear_batch(N, n, p, STEPS, seed=1234)
    xs, Ps = np.zeros((STEPS, N, n)), np.zeros((STEPS, N, n, n))
    for i in range(N):
        if kind == orc.INFORMATION:
            f = orc.Filter.information_from_state(d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
        else:
            f = orc.Filter.ldkf(kind, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
        for t in range(STEPS):
            assert f.update(d["y"][t, i]) == orc.OK
            xs[t, i], Ps[t, i] = f.state(), f.covariance()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x_steps=xs[[0, 9, 49]], P_steps=Ps[[0, 9, 49]], steps=np.array([0, 9, 49]), **d)


def nldkf(kind, name, n, p, ekf=False):
    rng = np.random.default_rng(77)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    T = 10
    Phi = np.eye(n) + 1e-2 * rng.standard_normal((T, N, n, n))
    Ht = rng.standard_normal((T, N, p, n))
    real = rng.standard_normal((T, N, p)); comp = real + 1e-2 * rng.standard_normal((T, N, p))
    xs, Ps = np.zeros((N, n)), np.zeros((N, n, n))
    for i in range(N):
        f = orc.Filter.srif(x0[i], P0[i], R[i], p) if kind == orc.SRIF else orc.Filter.hybrid(x0[i], P0[i], None, R[i], p)
        if ekf:
            f.enable_ekf(True)
        for t in range(T):
            f.prepare(Phi[t, i], Ht[t, i])
            assert f.update_nl(real[t, i], comp[t, i]) == orc.OK
        xs[i], Ps[i] = f.state(), f.covariance()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x0=x0, P0=P0, R=R, Phi=Phi, Ht=Ht, real=real, comp=comp, x_final=xs, P_final=Ps)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    ldkf(orc.VANILLA, "vanilla_6x3", 6, 3)
    ldkf(orc.SQUAREROOT, "squareroot_6x3", 6, 3)
    ldkf(orc.INFORMATION, "information_6x3", 6, 3)
    nldkf(orc.SRIF, "srif_12x6", 12, 6)
    nldkf(orc.HYBRID, "hybrid_ckf_6x2", 6, 2, ekf=False)
    nldkf(orc.HYBRID, "hybrid_ekf_6x2", 6, 2, ekf=True)
    print("wrote", sorted(os.listdir(OUT)))
