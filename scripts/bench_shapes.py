"""Throughput of Vanilla shapes that have no exact register kernel: the padded register kernels (kb_vanilla_pad.hip)
against the run-time-dimension scratch kernel (forced with KB_FLAG_STRICT_SYMCHECK).  usage: python scripts/bench_shapes.py"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k

N = 1 << 20
for (n, p) in [(5, 2), (3, 1), (7, 3), (8, 4)]:
    rng = np.random.default_rng(n)
    F = np.eye(n) + 0.05 * rng.standard_normal((n, n)); H = rng.standard_normal((p, n))
    Q = 1e-3 * np.eye(n); R = 1e-2 * np.eye(p)
    res = {}
    for name, flags in (("padded_register", 0), ("generic_scratch", k.FLAG_STRICT_SYMCHECK)):
        b = ga.FilterBatch.new_ldkf(k.VANILLA, np.zeros(n), np.eye(n), F, None, H, Q, R, nfilters=N, flags=flags)
        y = torch.randn((p, N), dtype=torch.float64, device="cuda")
        s = torch.cuda.ExternalStream(b.stream())
        for _ in range(3):
            b.update_dev(y.data_ptr(), N)
        b.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        K = 20
        for _ in range(K):
            b.update_dev(y.data_ptr(), N)
        e1.record(s)
        b.synchronize()
        res[name] = e0.elapsed_time(e1) / K
    bytes_per = 8 * (4 * n * n + p * n + p * p + 2 * n + p)
    print(json.dumps({"shape": [n, p], "ms": res, "speedup": res["generic_scratch"] / res["padded_register"],
                      "padded_frac_of_8TBps": N * bytes_per / (res["padded_register"] * 1e-3) / 8e12}))

# SquareRoot / Information: padded instantiations cover n <= 6, p <= 4; 7/2 shows the scratch kernel beside them
for kind, name in ((k.SQUAREROOT, "squareroot"), (k.INFORMATION, "information")):
    for (n, p) in [(5, 2), (3, 1), (6, 4), (7, 2)]:
        rng = np.random.default_rng(n)
        F = np.eye(n) + 0.05 * rng.standard_normal((n, n)); H = rng.standard_normal((p, n))
        Q = 1e-3 * np.eye(n); R = 1e-2 * np.eye(p)
        b = ga.FilterBatch.new_ldkf(kind, np.zeros(n), np.eye(n), F, None, H, Q, R, nfilters=N,
                                    flags=k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0)
        y = torch.randn((p, N), dtype=torch.float64, device="cuda")
        s = torch.cuda.ExternalStream(b.stream())
        for _ in range(3):
            b.update_dev(y.data_ptr(), N)
        b.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        K = 10
        for _ in range(K):
            b.update_dev(y.data_ptr(), N)
        e1.record(s)
        b.synchronize()
        print(json.dumps({"kind": name, "shape": [n, p], "ms": e0.elapsed_time(e1) / K}))
