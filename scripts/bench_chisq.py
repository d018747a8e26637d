"""NewChiSquare (chisquare.go:16-95) throughput: statOD5044 closed-loop model (examples/statOD5044/main.go:36-57), truth +
Vanilla filter + NIS / NEES per run and step in one launch.  usage: python scripts/bench_chisq.py [runs]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "examples")
import gokalman_amd as ga
from gokalman_amd import _capi as k
import statod5044 as m

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
zero_u = np.zeros((1, 2))
truth = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, m.x0, m.P0, m.Fcl, m.Gcl, m.H, m.Q, m.R, nfilters=runs, noise=k.NOISE_AWGN, seed=5044)
kf = ga.FilterBatch.new_ldkf(k.VANILLA, m.x0, m.P0, m.Fcl, m.Gcl, m.H, m.Q, m.R, nfilters=runs)
ga.new_chi_square(kf, truth, zero_u, steps=600)     # warm-up, ~30 ms: the GPU clocks are back up (bench.py warm_clocks)
t0 = time.perf_counter()
nis, nees = ga.new_chi_square(kf, truth, zero_u, steps=m.SAMPLES)
dt_s = time.perf_counter() - t0
print(json.dumps({"config": "chi-square, statOD5044 closed loop n=4 p=2", "runs": runs, "steps": m.SAMPLES, "seconds": dt_s,
                  "run_steps_per_s": runs * m.SAMPLES / dt_s, "nis_mean": float(nis.mean()), "nees_mean": float(nees.mean())}))
