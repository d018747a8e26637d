"""Large-batch check of the other kinds (2^22 filters, shared model, per-filter x0): the last 4096 filters must bit-equal a
4096-filter batch fed the same data (persistent-grid SRIF pipeline: 128 tiles per workgroup).  usage: python scripts/soak_kinds.py"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth

N, small = 1 << 22, 4096
rng = np.random.default_rng(3)
d = synth.linear_batch(1, 6, 3, 1)
for kind, name in ((k.SQUAREROOT, "squareroot"), (k.INFORMATION, "information")):
    x0 = rng.standard_normal((N, 6))
    fl = k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0
    mk = lambda xs, n_: ga.FilterBatch.new_ldkf(kind, xs, d["P0"][0], d["F"][0], None, d["H"][0], d["Q"][0], d["R"][0], nfilters=n_, flags=fl)
    big, ref = mk(x0, N), mk(x0[-small:], small)
    y = torch.randn((3, N), dtype=torch.float64, device="cuda"); ys = y[:, -small:].contiguous()
    for _ in range(4):
        big.update_dev(y.data_ptr(), N); ref.update_dev(ys.data_ptr(), small)
    big.synchronize(); ref.synchronize()
    ok = np.array_equal(big.get(k.RAW_VEC, N - small, small), ref.get(k.RAW_VEC)) and np.array_equal(big.get(k.RAW_MAT, N - small, small), ref.get(k.RAW_MAT))
    print(json.dumps({"kind": name, "filters": N, "tail_bitwise_equal": bool(ok), "errors": int(np.count_nonzero(big.status(N - small, small)))}))
    del big, ref
for kind, name, n, p, dt, tdt in ((k.HYBRID, "hybrid", 6, 2, k.F64, torch.float64), (k.SRIF, "srif f32", 12, 6, k.F32, torch.float32)):
    x0 = rng.standard_normal((N, n)); P0 = np.diag(np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)]))
    R = np.diag(np.full(p, 1e-3))
    def mk(xs, n_):
        b = ga.FilterBatch(kind, n, p, 0, n_, dtype=dt)
        b.set(k.X, xs, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
        return b
    big, ref = mk(x0, N), mk(x0[-small:], small)
    Phi = (torch.eye(n, dtype=tdt, device="cuda").reshape(n * n, 1) + 1e-2 * torch.randn(n * n, N, dtype=tdt, device="cuda")).contiguous()
    Ht = torch.randn(p * n, N, dtype=tdt, device="cuda")
    real = torch.randn(p, N, dtype=tdt, device="cuda"); comp = real + 1e-2 * torch.randn(p, N, dtype=tdt, device="cuda")
    sub = lambda t: t[:, -small:].contiguous()
    Phis, Hts, reals, comps = sub(Phi), sub(Ht), sub(real), sub(comp)
    for _ in range(4):
        for b_, a_, h_, r_, c_, n_ in ((big, Phi, Ht, real, comp, N), (ref, Phis, Hts, reals, comps, small)):
            k.check(k.lib().kb_prepare_dev(b_._h, a_.data_ptr(), h_.data_ptr(), n_))
            k.check(k.lib().kb_update_nl_dev(b_._h, r_.data_ptr(), c_.data_ptr(), n_))
    big.synchronize(); ref.synchronize()
    ok = np.array_equal(big.get(k.RAW_VEC, N - small, small), ref.get(k.RAW_VEC)) and np.array_equal(big.get(k.RAW_MAT, N - small, small), ref.get(k.RAW_MAT))
    print(json.dumps({"kind": name, "filters": N, "tail_bitwise_equal": bool(ok), "errors": int(np.count_nonzero(big.status(N - small, small)))}))
    del big, ref
