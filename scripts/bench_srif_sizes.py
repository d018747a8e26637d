"""config E (SRIF 12/6 fp32, zero-copy Phi / Htilde) per-filter time against the batch size: how much of the 256k-filter figure is the
first generation of waves starting in lockstep (2048 resident two-lane waves = 64k filters per generation)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokalman_amd as ga  # noqa: E402
from gokalman_amd import _capi as k  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20]
sn, sp = 12, 6
for M in sizes:
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal((M, sn))
    P0 = np.zeros((M, sn, sn)); P0[:, np.arange(sn), np.arange(sn)] = [10.0] * 6 + [1.0] * 6
    R = np.zeros((M, sp, sp)); R[:, np.arange(sp), np.arange(sp)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(M, sp)))
    sb = ga.FilterBatch(k.SRIF, sn, sp, 0, M, dtype=k.F32)
    sb.set(k.X, x0, 1); sb.set(k.P, P0, 2); sb.set(k.R, R, 2, p_rows=sp); sb.init()
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    Phi = (torch.eye(sn, dtype=torch.float32, device="cuda").reshape(sn * sn, 1) + 1e-2 * torch.randn(sn * sn, M, dtype=torch.float32, device="cuda", generator=g)).contiguous()
    Ht = torch.randn(sp * sn, M, dtype=torch.float32, device="cuda", generator=g)
    real = torch.randn(sp, M, dtype=torch.float32, device="cuda", generator=g)
    comp = real + 1e-2 * torch.randn(sp, M, dtype=torch.float32, device="cuda", generator=g)
    torch.cuda.synchronize()
    s = torch.cuda.ExternalStream(sb.stream())

    def step():
        k.check(k.lib().kb_prepare_dev(sb._h, Phi.data_ptr(), Ht.data_ptr(), M))
        k.check(k.lib().kb_update_nl_dev(sb._h, real.data_ptr(), comp.data_ptr(), M))
    for _ in range(300):
        step()
    sb.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(100):
            step()
        e1.record(s)
        sb.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    print("SRIF 12/6 fp32 %8d filters: %7.1f us per step = %.3f ns per filter = %.3f of 8 TB/s on 1742 B per filter; kernel %s; errors %d"
          % (M, best, best * 1e3 / M, 1742.0 * M / (best * 1e-6) / 8e12, sb.last_kernel(), int(np.count_nonzero(sb.status()))), flush=True)
    del sb, Phi, Ht, real, comp
    torch.cuda.empty_cache()
