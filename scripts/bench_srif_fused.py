"""config E time-fused (kb_update_nl_steps_dev, T = 20 steps per launch, distinct Phi / Htilde / observations per step) against T single steps."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokalman_amd as ga  # noqa: E402
from gokalman_amd import _capi as k  # noqa: E402

M, T, n, p = 1 << 18, int(sys.argv[1]) if len(sys.argv) > 1 else 20, 12, 6
rng = np.random.default_rng(5)
x0 = rng.standard_normal((M, n))
P0 = np.zeros((M, n, n)); P0[:, np.arange(n), np.arange(n)] = [10.0] * 6 + [1.0] * 6
R = np.zeros((M, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(M, p)))
g = torch.Generator(device="cuda"); g.manual_seed(11)
Phi = (torch.eye(n, dtype=torch.float32, device="cuda").reshape(1, n * n, 1) + 1e-2 * torch.randn(T, n * n, M, dtype=torch.float32, device="cuda", generator=g)).contiguous()
Ht = torch.randn(T, p * n, M, dtype=torch.float32, device="cuda", generator=g)
real = torch.randn(T, p, M, dtype=torch.float32, device="cuda", generator=g)
comp = real + 1e-2 * torch.randn(T, p, M, dtype=torch.float32, device="cuda", generator=g)
torch.cuda.synchronize()
for fused in (False, True, False, True):
    sb = ga.FilterBatch(k.SRIF, n, p, 0, M, dtype=k.F32)
    sb.set(k.X, x0, 1); sb.set(k.P, P0, 2); sb.set(k.R, R, 2, p_rows=p); sb.init()
    s = torch.cuda.ExternalStream(sb.stream())

    def run():
        if fused:
            sb.update_nl_steps_dev(Phi.data_ptr(), Ht.data_ptr(), M, n * n * M, p * n * M, real.data_ptr(), comp.data_ptr(), M, p * M, T)
        else:
            for t in range(T):
                k.check(k.lib().kb_prepare_dev(sb._h, Phi[t].data_ptr(), Ht[t].data_ptr(), M))
                k.check(k.lib().kb_update_nl_dev(sb._h, real[t].data_ptr(), comp[t].data_ptr(), M))
    for _ in range(15):
        run()
    sb.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10):
        run()
    e1.record(s)
    sb.synchronize()
    us = e0.elapsed_time(e1) / (10 * T) * 1e3
    print("SRIF 12/6 fp32, %d filters, %s: %.1f us per step = %.2f G filter-steps/s; kernel %s; errors %d"
          % (M, "ONE launch of %d steps" % T if fused else "%d single steps" % T, us, M / us / 1e3, sb.last_kernel(), int(np.count_nonzero(sb.status()))), flush=True)
    del sb
