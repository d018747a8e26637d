"""Throughput of the split-lane kernels (one filter over 4 / 8 lanes: kb_vanilla_split.h, kb_squareroot_split.h,
kb_information_split.h) across the shapes between 7 and 16 states, 262 144 filters, fp64: microseconds per step, filter-steps/s
and the fraction of 8 TB/s the packed bytes of a step (state in + out, F, H, Q, R, y) amount to.
usage: python scripts/bench_split_shapes.py [--n=N] [--kind=vanilla,squareroot,information] [--max-n=8] [--full] [--awgn]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k

N = 1 << 18
KINDS, MAXN, MINN, FULL, AWGN = None, 16, 0, False, False
for a in sys.argv[1:]:
    if a.startswith("--n="):
        N = int(a[4:])
    if a.startswith("--kind="):
        KINDS = a[7:].split(",")
    if a == "--full":
        FULL = True
    if a == "--awgn":
        AWGN = True
    if a.startswith("--min-n="):
        MINN = int(a[8:])
    if a.startswith("--max-n="):
        MAXN = int(a[8:])
tri = lambda n: n * (n + 1) // 2
SHAPES = [(7, 3), (8, 2), (8, 4), (9, 3), (10, 5), (12, 6), (12, 3), (12, 8), (14, 7), (16, 4), (16, 8)]
for kind, name in ((k.VANILLA, "vanilla"), (k.SQUAREROOT, "squareroot"), (k.INFORMATION, "information")):
    if KINDS is not None and name not in KINDS:
        continue
    for (n, p) in SHAPES:
        if n > MAXN or n < MINN:
            continue
        rng = np.random.default_rng(n)
        # per-filter models (every filter reads its own F, H, Q, R from HBM): one base model, scaled per filter
        sc = (1.0 + 0.01 * rng.random(N))[:, None, None]
        F = np.eye(n) + sc * (0.05 * rng.standard_normal((n, n))); H = sc * rng.standard_normal((p, n))
        Q = sc * (1e-3 * np.eye(n)); R = sc * (1e-2 * np.eye(p))
        x0 = np.zeros((N, n)); P0 = np.broadcast_to(np.eye(n), (N, n, n))
        b = ga.FilterBatch.new_ldkf(kind, x0, P0, F, None, H, Q, R,
                                    flags=(k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0) | (k.FLAG_FULL_ESTIMATE if FULL else 0),
                                    noise=k.NOISE_AWGN if AWGN else k.NOISE_NOISELESS, seed=7)
        del F, Q
        y = torch.randn((p, N), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
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
        us = e0.elapsed_time(e1) / K * 1e3
        packed = 8 * (2 * (n + tri(n)) + n * n + p * n + tri(n) + tri(p) + p)
        print(json.dumps({"kind": name, "shape": [n, p], "us": round(us, 1), "filter_steps_per_s": round(N / (us * 1e-6) / 1e9, 3),
                          "packed_B": packed, "frac_of_8TBps": round(N * packed / (us * 1e-6) / 8e12, 3), "full": FULL, "awgn": AWGN, "errors": int(b.status().any())}), flush=True)
