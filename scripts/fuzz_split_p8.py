"""Differential fuzz of the Vanilla split kernels at 7, 8 measurements (S^-1 once per filter: kb_vanilla_split.h dist_inverse) against the oracle
on STRUCTURED innovation covariances: measurement rows that are zero, duplicated, or scaled over sixteen decades; R diagonal with equal entries
(ties in the pivot search), with a huge dynamic range, or dense.  Per case: the filters whose status differs from the oracle's return code,
and the worst relative error over the filters both sides accept.  usage: python scripts/fuzz_split_p8.py [cases]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k
from oracle import oracle as orc

N, steps = 192, 3


def run_case(case):
    rng = np.random.default_rng(9000 + case)
    n, p = [(12, 8), (16, 8), (14, 7), (11, 7), (9, 8), (13, 8)][case % 6]
    F = np.eye(n) + 0.05 * rng.standard_normal((N, n, n))
    H = rng.standard_normal((N, p, n))
    A = rng.standard_normal((N, n, n)); Q = 1e-3 * np.einsum("nij,nkj->nik", A, A) + 1e-4 * np.eye(n)
    x0 = rng.standard_normal((N, n)); P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = rng.uniform(1.0, 10.0, size=(N, n))
    kindR = case % 4
    if kindR == 0:
        R = np.broadcast_to(0.5 * np.eye(p), (N, p, p)).copy()                                   # equal diagonal: ties
    elif kindR == 1:
        R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = 10.0 ** rng.uniform(-8, 8, size=(N, p))   # sixteen decades
    elif kindR == 2:
        B = rng.standard_normal((N, p, p)); R = np.einsum("nij,nkj->nik", B, B) + 1e-6 * np.eye(p)          # dense
    else:
        R = np.zeros((N, p, p))                                                                  # none: S = H P- H^T alone
    mode = (case // 4) % 3
    sel = rng.random(N) < 0.5
    if mode == 0:
        H[sel, rng.integers(0, p)] = 0.0                      # a zero measurement row in half of the filters
    elif mode == 1:
        H[sel, p - 1] = H[sel, 1]                             # a duplicated row
    else:
        H *= (10.0 ** rng.uniform(-4, 4, size=(N, p)))[:, :, None]   # rows scaled over eight decades
    y = rng.standard_normal((steps, N, p))
    b = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, None, H, Q, R)
    fs = [orc.Filter.ldkf(orc.VANILLA, x0[i], P0[i], F[i], None, H[i], Q[i], R[i]) for i in range(N)]
    mism, worst = 0, 0.0
    alive = np.ones(N, dtype=bool)
    for t in range(steps):
        b.update(y[t])
        st = b.status()
        for i, f in enumerate(fs):
            rc = f.update(y[t, i])
            if (rc != orc.OK) != bool(st[i]):
                mism += 1; alive[i] = False
            if rc != orc.OK:
                alive[i] = False
    xs, Ps = b.get(k.STATE), b.get(k.COVAR)
    for i in np.nonzero(alive)[0]:
        worst = max(worst, float(np.linalg.norm(xs[i] - fs[i].state()) / max(np.linalg.norm(fs[i].state()), 1e-300)),
                    float(np.linalg.norm(Ps[i] - fs[i].covariance()) / max(np.linalg.norm(fs[i].covariance()), 1e-300)))
    return {"case": case, "shape": [n, p], "R": ["equal diagonal", "sixteen decades", "dense", "zero"][kindR], "H": ["zero row", "duplicated row", "scaled rows"][mode],
            "status_mismatches": mism, "filters_compared": int(alive.sum()), "filters_failed_on_both_sides": int(N - alive.sum() - mism), "worst_rel_error": worst}


if __name__ == "__main__":
    CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    tot_mismatch, worst_all = 0, 0.0
    for case in range(CASES):
        r = run_case(case)
        tot_mismatch += r["status_mismatches"]; worst_all = max(worst_all, r["worst_rel_error"])
        print(json.dumps(r), flush=True)
    print(json.dumps({"cases": CASES, "status_mismatches_total": tot_mismatch, "worst_rel_error": worst_all}))
