"""64-bit addressing check: 2^24 Vanilla 6/3 filters (model block 10.9 GB, past every 32-bit byte offset), shared model,
per-filter x0; the last 4096 filters must equal a 4096-filter batch fed the same data.  usage: python scripts/soak_large.py [log2N]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
N, small = 1 << lg, 4096
d = synth.linear_batch(1, 6, 3, 2)
rng = np.random.default_rng(1)
x0 = rng.standard_normal((N, 6))
big = ga.FilterBatch.new_ldkf(k.VANILLA, x0, d["P0"][0], d["F"][0], None, d["H"][0], d["Q"][0], d["R"][0], nfilters=N)
ref = ga.FilterBatch.new_ldkf(k.VANILLA, x0[-small:], d["P0"][0], d["F"][0], None, d["H"][0], d["Q"][0], d["R"][0], nfilters=small)
y = torch.randn((3, N), dtype=torch.float64, device="cuda")
ys = y[:, -small:].contiguous()
t0 = time.perf_counter()
for _ in range(5):
    big.update_dev(y.data_ptr(), N)
    ref.update_dev(ys.data_ptr(), small)
big.synchronize(); ref.synchronize()
dt = time.perf_counter() - t0
ok = np.array_equal(big.get(k.STATE, N - small, small), ref.get(k.STATE)) and np.array_equal(big.get(k.COVAR, N - small, small), ref.get(k.COVAR))
first_ok = np.all(np.isfinite(big.get(k.STATE, 0, 8)))
print(json.dumps({"filters": N, "device_GB": torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9,
                  "tail_bitwise_equal": bool(ok), "head_finite": bool(first_ok), "errors": int(np.count_nonzero(big.status(N - small, small))),
                  "ms_per_step": dt / 5 * 1e3}))
