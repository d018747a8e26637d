"""Vanilla + BatchNoise (noise.go:67-106) per-step time at 256k filters, per-filter models: 12/6 (four lanes, exact, NOISET = 2), 16/8 and
10/4 (run-time-everything split kernels), 6/3 (register kernel).  VERDICT r05 task 5: 12/6 was 8.7 ms on the statement kernel."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokalman_amd as ga  # noqa: E402
from gokalman_amd import _capi as k, synth  # noqa: E402

M, K = 1 << 18, 40
for n, p in ((12, 6), (16, 8), (10, 4), (6, 3)):
    d = synth.linear_batch(M, n, p, 1, seed=11)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], np.zeros((n, n)), np.zeros((p, p)), nfilters=M)
    rng = np.random.default_rng(1)
    b.set_batch_noise(1e-2 * rng.standard_normal((K + 20, n)), 1e-2 * rng.standard_normal((K + 20, p)))
    y = torch.from_numpy(np.ascontiguousarray(d["y"][0].T)).cuda()
    torch.cuda.synchronize()
    s = torch.cuda.ExternalStream(b.stream())
    for _ in range(10):
        b.update_dev(y.data_ptr(), M)
    b.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(K):
        b.update_dev(y.data_ptr(), M)
    e1.record(s)
    b.synchronize()
    print("Vanilla %d/%d + BatchNoise, %d filters: %.1f us per step; kernel %s; filters with an error status %d (S is exactly singular past n measurements)"
          % (n, p, M, e0.elapsed_time(e1) / K * 1e3, b.last_kernel(), int(np.count_nonzero(b.status()))))
