"""Where a BatchNoise run (zero Q / R, noise.go:89-98) leaves the oracle, step by step: default dispatch against
KB_FLAG_STATEMENT_KERNELS, n/p given on the command line (diagnostic for tests/test_vanilla_split_gpu.py)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
from oracle import oracle as orc
from tests.test_vanilla_split_gpu import _model

for n, p in [(12, 3), (9, 2), (15, 4), (6, 3), (8, 4)]:
    N, steps = 100, 8
    d = _model(N, n, p, 0, steps, 900 + n)
    rng = np.random.default_rng(n)
    proc, meas = 1e-2 * rng.standard_normal((steps, n)), 1e-2 * rng.standard_normal((steps, p))
    ZQ, ZR = np.zeros((n, n)), np.zeros((p, p))
    for name, fl in (("default", 0), ("statement", k.FLAG_STATEMENT_KERNELS)):
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], ZQ, ZR, nfilters=N, flags=k.FLAG_FULL_ESTIMATE | fl)
        b.set_batch_noise(proc, meas)
        fs = [orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], ZQ, ZR) for i in range(N)]
        for t in range(steps):
            est = b.update(d["y"][t])
            rcs = np.array([f.update(d["y"][t, i], None, proc[t], meas[t], proc[t]) for i, f in enumerate(fs)])
            st = b.status(); b.clear_status()
            xs, Ps = np.array([f.state() for f in fs]), np.array([f.covariance() for f in fs])
            ex = np.linalg.norm(est.state() - xs, axis=1) / np.linalg.norm(xs, axis=1)
            eP = np.abs(est.covariance() - Ps).reshape(N, -1).max(axis=1)
            print("%2d/%d %-9s step %d: oracle fails %3d, engine fails %3d (same filters: %s); x rel err max %.2e median %.2e; |dP| max %.2e; |P| oracle max %.2e"
                  % (n, p, name, t, (rcs != 0).sum(), (st != 0).sum(), np.array_equal(rcs != 0, st != 0), ex.max(), np.median(ex), eP.max(), np.abs(Ps).max()))
