"""Diagnostic: per-call latency of the drop-in (N = 1, host vectors) path: kb_update alone, kb_update + the one-call
Estimate snapshot (kb_get_estimate: every member, one synchronisation), the round-1 pattern (update + two getters), and
the CPU oracle for scale."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
from oracle import oracle as orc  # diagnostic only
d = synth.linear_batch(1, 6, 3, 1)
for flags, name in ((0, "state-only"), (k.FLAG_FULL_ESTIMATE, "full estimate")):
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=flags)
    y = d["y"][0]
    for _ in range(50): b.update(y)
    K = 2000
    t = time.perf_counter()
    for _ in range(K): b.update(y, snapshot=False)
    dt = (time.perf_counter() - t) / K
    t = time.perf_counter()
    for _ in range(K): b.update(y, snapshot=True)
    dt1 = (time.perf_counter() - t) / K
    t = time.perf_counter()
    for _ in range(200): b.update(y, snapshot=False); b.get(k.STATE); b.get(k.COVAR)
    dt2 = (time.perf_counter() - t) / 200
    print("N=1 %s: kb_update %.1f us/call; update + owning Estimate (kb_get_estimate) %.1f us; update + State() + Covariance() getters %.1f us"
          % (name, dt * 1e6, dt1 * 1e6, dt2 * 1e6))
f = orc.Filter.ldkf(orc.VANILLA, d["x0"][0], d["P0"][0], d["F"][0], None, d["H"][0], d["Q"][0], d["R"][0])
t = time.perf_counter()
for _ in range(2000): f.update(d["y"][0, 0])
print("CPU oracle (ctypes): %.1f us/call" % ((time.perf_counter() - t) / 2000 * 1e6))
