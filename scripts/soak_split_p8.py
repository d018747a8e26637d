"""64-bit addressing check of the Vanilla split kernels at 7, 8 measurements (S^-1 once per filter: kb_vanilla_split.h dist_inverse):
2^20 filters of 12/8, 14/7 and 16/8 with per-filter models (model blocks of 2.9 - 4.7 GB: past every 32-bit byte offset) -- the LAST
4096 filters must equal a 4096-filter batch fed the same data bit for bit, and 64 sampled filters the oracle at 1e-9.
usage: python scripts/soak_split_p8.py [log2N]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k
from oracle import oracle as orc

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, small, steps = 1 << lg, 4096, 3
for (n, p) in ((12, 8), (14, 7), (16, 8)):
    rng = np.random.default_rng(100 * n + p)
    sc = (1.0 + 0.01 * rng.random(N))[:, None, None]
    F = np.eye(n) + sc * (0.05 * rng.standard_normal((n, n))); H = sc * rng.standard_normal((p, n))
    Q = sc * (1e-3 * np.eye(n)); R = sc * (1e-2 * (np.eye(p) + 0.5 * np.ones((p, p))))
    x0 = rng.standard_normal((N, n)); P0 = np.broadcast_to(np.eye(n), (N, n, n))
    big = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, None, H, Q, R)
    ref = ga.FilterBatch.new_ldkf(k.VANILLA, x0[-small:], P0[-small:], F[-small:], None, H[-small:], Q[-small:], R[-small:])
    y = torch.randn((steps, p, N), dtype=torch.float64, device="cuda")
    ys = y[:, :, -small:].contiguous()   # (kept alive: the launches are asynchronous)
    for t in range(steps):
        big.update_dev(y[t].data_ptr(), N)
        ref.update_dev(ys[t].data_ptr(), small)
    big.synchronize(); ref.synchronize()
    same = bool(np.array_equal(big.get(k.STATE, N - small, small), ref.get(k.STATE)) and np.array_equal(big.get(k.COVAR, N - small, small), ref.get(k.COVAR)))
    yh = y.cpu().numpy()
    worst = 0.0
    for i in list(rng.integers(0, N, size=60)) + [0, 1, N - 2, N - 1]:
        f = orc.Filter.ldkf(orc.VANILLA, x0[i], P0[i], F[i], None, H[i], Q[i], R[i])
        for t in range(steps):
            assert f.update(yh[t, :, i]) == orc.OK
        xs, Ps = big.get(k.STATE, int(i), 1)[0], big.get(k.COVAR, int(i), 1)[0]
        worst = max(worst, float(np.linalg.norm(xs - f.state()) / np.linalg.norm(f.state())), float(np.linalg.norm(Ps - f.covariance()) / np.linalg.norm(f.covariance())))
    print(json.dumps({"shape": [n, p], "filters": N, "model_block_GB": round(N * 8 * (n * n + p * n + n * (n + 1) // 2 + p * (p + 1) // 2) / 1e9, 2),
                      "last_4096_bit_equal_to_a_small_batch": same, "errors": int(np.count_nonzero(big.status())), "worst_rel_error_vs_oracle_64_filters": worst}), flush=True)
    del big, ref, y, F, H, Q, R
