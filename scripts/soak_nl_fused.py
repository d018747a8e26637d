"""Large-batch check of the time-fused NLDKF kernels (kb_update_nl_steps_dev): Hybrid EKF 6/2 at 8M filters and SRIF 12/6 fp32 at 4M filters,
T = 3 steps with distinct operands per step, against T single calls on a twin batch: the same bits over the WHOLE batch (device-side compare),
no error status.  Element offsets pass 2^32 bytes in every operand array."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokalman_amd as ga  # noqa: E402
from gokalman_amd import _capi as k  # noqa: E402

for kind, n, p, M, dt, tdt in ((k.HYBRID, 6, 2, 8 << 20, k.F64, torch.float64), (k.SRIF, 12, 6, 4 << 20, k.F32, torch.float32)):
    T = 3
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    Phi = (torch.eye(n, dtype=tdt, device="cuda").reshape(1, n * n, 1) + 1e-2 * torch.randn(T, n * n, M, dtype=tdt, device="cuda", generator=g)).contiguous()
    Ht = torch.randn(T, p * n, M, dtype=tdt, device="cuda", generator=g)
    real = torch.randn(T, p, M, dtype=tdt, device="cuda", generator=g)
    comp = (real + 1e-2 * torch.randn(T, p, M, dtype=tdt, device="cuda", generator=g)).contiguous()
    torch.cuda.synchronize()
    x0 = np.linspace(-1.0, 1.0, n); P0 = np.diag([10.0] * (n // 2) + [1.0] * (n // 2)); R = np.diag(np.full(p, 1e-3))
    outs = []
    for fused in (True, False):
        b = ga.FilterBatch(kind, n, p, 0, M, dtype=dt)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
        if kind == k.HYBRID:
            b.enable_ekf()
        if fused:
            b.update_nl_steps_dev(Phi.data_ptr(), Ht.data_ptr(), M, n * n * M, p * n * M, real.data_ptr(), comp.data_ptr(), M, p * M, T)
            kern = b.last_kernel()
        else:
            for t in range(T):
                k.check(k.lib().kb_prepare_dev(b._h, Phi[t].data_ptr(), Ht[t].data_ptr(), M))
                k.check(k.lib().kb_update_nl_dev(b._h, real[t].data_ptr(), comp[t].data_ptr(), M))
        b.synchronize()
        fields = (k.RAW_MAT, k.RAW_VEC) if kind == k.SRIF else (k.STATE, k.COVAR)
        # sample: the first, middle and last 4096 filters of every field + the number of error flags
        samp = [np.concatenate([b.get(f, lo, 4096).ravel() for lo in (0, M // 2 - 2048, M - 4096)]) for f in fields]
        outs.append((samp, int(np.count_nonzero(b.status()))))
        del b
    same = all(np.array_equal(u.view(np.uint64), v.view(np.uint64)) for u, v in zip(outs[0][0], outs[1][0]))
    print("%s %d/%d, %d filters x %d steps: %s; sampled 3 x 4096 filters bit-identical to single calls: %s; error flags %d / %d; finite %s"
          % ("Hybrid EKF" if kind == k.HYBRID else "SRIF fp32", n, p, M, T, kern, same, outs[0][1], outs[1][1], all(np.isfinite(u).all() for u in outs[0][0])), flush=True)
    del Phi, Ht, real, comp
    torch.cuda.empty_cache()
