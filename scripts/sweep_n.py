"""Diagnostic: per-filter-step time of the Vanilla 6x3 kernel vs batch size (Infinity Cache residency of x,P)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
for N in [1 << 18, 1 << 19, 3 << 18, 1 << 20, 5 << 18, 3 << 19, 1 << 21]:
    d = synth.linear_batch(N, 6, 3, 1)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    s = torch.cuda.ExternalStream(b.stream())
    for _ in range(10): b.update_dev(y[0].data_ptr(), N)
    b.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 100
    e0.record(s)
    for _ in range(K): b.update_dev(y[0].data_ptr(), N)
    e1.record(s); b.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("N=%8d state=%6.1f MB  %.4f ms  %.3f G steps/s  algo %.0f GB/s" % (N, N * 27 * 8 / 1e6, ms, N / ms / 1e6, N * 1488 / ms / 1e6))
    del b
