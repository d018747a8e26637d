"""Time-fused SquareRoot 6/3 (kb_update_steps_dev) against T one-step launches, bit for bit: how many of 1000 filters differ after T steps.
Diagnostic for the variants of scripts/ab_libs.sh (-DKB_SQRT_FUSED_FASTDIV=0 with and without -ffp-contract=off ...)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokalman_amd as ga  # noqa: E402
from gokalman_amd import _capi as k, synth  # noqa: E402

N = 1000
for T in (1, 2, 6):
    d = synth.linear_batch(N, 6, 3, T, seed=synth.SEED + 79)
    b = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    b1 = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    torch.cuda.synchronize()
    b.update_steps_dev(y.data_ptr(), N, T)
    for t in range(T):
        b1.update_dev(y[t].data_ptr(), N)
    b.synchronize(); b1.synchronize()
    xs, x1, Ps, P1 = b.get(k.STATE), b1.get(k.STATE), b.get(k.RAW_MAT), b1.get(k.RAW_MAT)
    diff = np.any(xs != x1, axis=1) | np.any(Ps.reshape(N, -1) != P1.reshape(N, -1), axis=1)
    rel = max(synth.rel_frobenius(xs, x1), synth.rel_frobenius(Ps, P1))
    print("T = %d: %d of %d filters differ from T one-step launches; max rel-Frobenius %.2e" % (T, int(diff.sum()), N, rel))
