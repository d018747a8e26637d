"""Diagnostic: PCIe-inclusive rate of the host-buffer path (kb_update with numpy measurements) at 1M filters."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
N = 1 << 20
d = synth.linear_batch(N, 6, 3, 2)
b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
for _ in range(3): b.update(d["y"][0])
t = time.perf_counter(); K = 20
for i in range(K): b.update(d["y"][i % 2])
dt = (time.perf_counter() - t) / K
print("host path: %.3f ms per 1M-filter step = %.2f G filter-steps/s (25 MB H2D + pack + step + sync per call)" % (dt * 1e3, N / dt / 1e9))
