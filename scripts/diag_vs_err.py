import sys; sys.path.insert(0, '.')
import numpy as np
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
from oracle import oracle as orc
for n, p in ((12, 6), (10, 4), (9, 3), (16, 4)):
    N, steps = 256, 20
    d = synth.linear_batch(N, n if n % 2 == 0 else n + 1, p, steps)
    if n % 2: d = {kk: (v[:, :n, :n] if kk in ("F", "P0", "Q") else v[:, :n] if kk == "x0" else v[:, :, :n] if kk == "H" else v) for kk, v in d.items()}
    a = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    for t in range(steps):
        a.update(d["y"][t])
    xo, Po, nerr = orc.ldkf_batch(orc.VANILLA, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
    print(n, p, a.last_kernel()[:60], "vs oracle: x %.2e P %.2e" % (synth.rel_frobenius(a.get(k.STATE), xo), synth.rel_frobenius(a.get(k.COVAR), Po)))
