#!/usr/bin/env python3
"""examples/robot of the reference (examples/robot/main.go) on the MI355X engine, call for call: a pure-predictor
Vanilla with AWGN noise (mcKF) and a Vanilla under test (chiKF), ONE filter each as in main.go:31-32;
NewMonteCarloRuns(sims, steps, 1, controls, mcKF) -> runs.AsCSV(headers) -> montecarlo-<header>.csv (every run, mean,
stddev per step, main.go:41-47); NewChiSquare(chiKF, runs, controls, true, true) -> chisquare.csv (main.go:49-59).
The reference runs 50 simulations one after the other; here they are one launch, and --runs may be much larger.
usage: python examples/robot.py [--runs N] [outdir]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokalman_amd as ga
from gokalman_amd import _capi as k

dt = 0.1
F = np.array([[1, dt], [0, 1]])
G = np.array([[0.5 * dt * dt], [dt]])
H = np.array([[1.0, 0]])
R = np.array([[0.05]])
Q = np.array([[5e-2, 5e-4], [5e-4, 1e-3]])   # "Q small" (main.go:22)
x0, P0 = np.zeros(2), 2.0 * np.eye(2)
STEPS = 120


def main(outdir, runs, seed=1):
    os.makedirs(outdir, exist_ok=True)
    mc_x0 = np.linalg.cholesky(P0) @ np.random.default_rng(seed).standard_normal(2)   # main.go:27-29
    controls = np.cos(0.75 * (np.arange(STEPS) + 1) * 0.1).reshape(STEPS, 1)             # main.go:36-38
    mckf = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, mc_x0, P0, F, G, H, Q, R, noise=k.NOISE_AWGN, seed=seed)   # main.go:31
    chikf = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, G, H, Q, R)                                              # main.go:32
    mc = ga.new_monte_carlo_runs(runs, STEPS, 1, controls, mckf)                                                   # main.go:41
    headers = ["xi", "xi_dot"]
    for i, contents in enumerate(mc.as_csv(headers)):                                                              # main.go:43-47
        with open(os.path.join(outdir, "montecarlo-%s.csv" % headers[i]), "w") as fh:
            fh.write(contents)
    nis, nees = ga.new_chi_square(chikf, mc, controls, True, True)                                                 # main.go:49
    with open(os.path.join(outdir, "chisquare.csv"), "w") as fh:
        fh.write("NIS,NEES\n")
        for s in range(STEPS):
            fh.write("%f,%f\n" % (nis[s], nees[s]))
    return {"nis_mean": float(nis.mean()), "nees_mean": float(nees.mean()), "mc": mc}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=50)   # sims := 50 (main.go:34)
    ap.add_argument("outdir", nargs="?", default="./robot_out")
    a = ap.parse_args()
    out = main(a.outdir, a.runs)
    print({kk: v for kk, v in out.items() if kk != "mc"})
