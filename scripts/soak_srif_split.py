"""kb_srif_split.h at a size where byte offsets pass 2^32: 2 097 152 filters x 16/6 fp64 (4.6 GB of state, 4.3 GB of Phi), zero-copy
Phi / Htilde, three Updates with a Predict() in between; the first and the last tile and 64 random filters against the oracle, the whole
batch for status words.  usage (inside a gpurun command): python scripts/soak_srif_split.py [log2 filters]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
from oracle import oracle as orc

N = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 21)
n, p, plan = 16, 6, "upuu"
g = torch.Generator(device="cuda"); g.manual_seed(5)
x0 = torch.randn(N, n, dtype=torch.float64, device="cuda", generator=g).cpu().numpy()
P0d = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
Rd = np.exp(np.random.default_rng(1).uniform(np.log(1e-4), np.log(1e-2), size=p))
b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=k.F64)
b.set(k.X, x0, 1); b.set(k.P, np.diag(P0d), 2); b.set(k.R, np.diag(Rd), 2, p_rows=p); b.init()
pick = np.unique(np.concatenate([np.arange(64), np.arange(N - 64, N), np.random.default_rng(2).integers(0, N, 64)]))
fs = {int(i): orc.Filter.srif(x0[i], np.diag(P0d), np.diag(Rd), p) for i in pick}
eye = torch.eye(n, dtype=torch.float64, device="cuda").reshape(n * n, 1)
t0 = time.perf_counter()
for t, what in enumerate(plan):
    Phi = (eye + 1e-2 * torch.randn(n * n, N, dtype=torch.float64, device="cuda", generator=g)).contiguous()
    Ht = torch.randn(p * n, N, dtype=torch.float64, device="cuda", generator=g)
    real = torch.randn(p, N, dtype=torch.float64, device="cuda", generator=g)
    comp = (real + 1e-2 * torch.randn(p, N, dtype=torch.float64, device="cuda", generator=g)).contiguous()
    torch.cuda.synchronize()
    k.check(k.lib().kb_prepare_dev(b._h, Phi.data_ptr(), Ht.data_ptr(), N))
    if what == "p":
        b.predict_nl(snapshot=False)
    else:
        k.check(k.lib().kb_update_nl_dev(b._h, real.data_ptr(), comp.data_ptr(), N))
    b.synchronize()
    idx = torch.from_numpy(pick).cuda()
    Ph = Phi[:, idx].cpu().numpy().T.reshape(-1, n, n); Hh = Ht[:, idx].cpu().numpy().T.reshape(-1, p, n)
    rh = real[:, idx].cpu().numpy().T; ch = comp[:, idx].cpu().numpy().T
    for j, i in enumerate(pick):
        f = fs[int(i)]
        f.prepare(Ph[j], Hh[j])
        assert (f.predict_nl() if what == "p" else f.update_nl(rh[j], ch[j])) == orc.OK
    del Phi, Ht, real, comp
worst = 0.0
for i in pick:
    f = fs[int(i)]
    R = b.get(k.RAW_MAT, int(i), 1)[0]; bv = b.get(k.RAW_VEC, int(i), 1)[0]
    worst = max(worst, np.linalg.norm(R - f.raw_mat()) / np.linalg.norm(f.raw_mat()), np.linalg.norm(bv - f.raw_vec()) / np.linalg.norm(f.raw_vec()))
nbad = int(np.count_nonzero(b.status()))
print("soak_srif_split: %d filters x %d/%d, %s: worst relative error over %d sampled filters %.2e, filters with a status bit %d, %.1f s"
      % (N, n, p, plan, len(pick), worst, nbad, time.perf_counter() - t0))
assert worst <= 1e-9 and nbad == 0 and b.step() == len(plan)
