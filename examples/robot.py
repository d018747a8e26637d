#!/usr/bin/env python3
"""examples/robot of the reference (examples/robot/main.go) on the MI355X engine: Monte-Carlo runs of a
2-state robot with a cosine control (montecarlo-*.csv: mean, stddev per step) and the NIS / NEES chi-square
test of a Vanilla filter (chisquare.csv).  usage: python examples/robot.py [--runs N] [outdir]"""
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
    mckf = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, mc_x0, P0, F, G, H, Q, R, nfilters=runs, noise=k.NOISE_AWGN, seed=seed)
    chikf = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, G, H, Q, R, nfilters=runs)
    mc = ga.new_monte_carlo_runs(runs, STEPS, 1, controls, mckf)
    for i, h in enumerate(["xi", "xi_dot"]):
        with open(os.path.join(outdir, "montecarlo-%s.csv" % h), "w") as fh:
            fh.write("%s-mean,%s-stddev\n" % (h, h))
            for s in range(STEPS):
                fh.write("%f,%f\n" % (mc.mean(s)[i], mc.stddev(s)[i]))
    nis, nees = ga.new_chi_square(chikf, mckf, STEPS, controls)
    with open(os.path.join(outdir, "chisquare.csv"), "w") as fh:
        fh.write("NIS,NEES\n")
        for s in range(STEPS):
            fh.write("%f,%f\n" % (nis[s], nees[s]))
    return {"nis_mean": float(nis.mean()), "nees_mean": float(nees.mean())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=4096)
    ap.add_argument("outdir", nargs="?", default="./robot_out")
    a = ap.parse_args()
    print(main(a.outdir, a.runs))
