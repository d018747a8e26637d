#!/usr/bin/env python3
"""examples/statOD5044 of the reference (examples/statOD5044/main.go) on the MI355X engine: Monte-Carlo
runs of the open- and closed-loop pure predictors (means / stddevs per step), a truth trajectory, the three
linear filters tracking it, and the NIS / NEES chi-square statistics.  The reference runs 15 Monte-Carlo
samples sequentially on ONE filter; here `--runs` copies of that filter run as one launch and AsCSV writes the reference's
per-run columns (mc-<ctrl|noctrl>-<header>.csv).   usage: python examples/statod5044.py [--runs N] [outdir]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd.exporter import CSVExporter

dt = 0.1
F = np.array([[1, 0.1, 0, 7.726e-2], [4.015e-7, 1, 0, 1.545], [-2.319e-16, -1.732e-9, 1, 0.1], [-6.956e-15, -3.465e-8, 0, 1]])
G = np.array([[5e-3, 3.85e-7], [0.1, 1.157e-5], [-5.775e-11, 7.487e-7], [1.732e-9, 1.498e-5]])
H = np.array([[1.0, 0, 0, 0], [0, 0, 1, 0]])
Q = np.array([[6.669e-16, 1.001e-14, 3.823e-19, 5.150e-18], [1.001e-14, 2.002e-13, 1.030e-17, 1.545e-16],
              [3.862e-19, 1.030e-17, 6.667e-19, 1.000e-17], [5.150e-18, 1.545e-16, 1.000e-17, 2.000e-16]])
R = np.diag([2e-3, 2e-5]) / dt
T = np.array([[0.930124736616832, 1.395260337125255, -0.000008568056356, 15.440297905873823],
              [0.000001749639349, 0.000000859493456, 0.001999922457941, 5.177881640687808]])
Fcl = F - G @ T                      # main.go:50-52
Gcl = np.zeros((4, 2))
x0, P0 = np.array([2, 0.5, 0, 0.0]), np.diag([5, 1, 0.01, 1e-5])
SAMPLES = int((5.431e3 / 50) / dt)   # main.go:66-67: 1086 steps


def main(outdir, runs):
    os.makedirs(outdir, exist_ok=True)
    headers = ["dr", "dr_dot", "dtheta", "dtheta_dot"]
    zero_u = np.zeros((1, 2))
    # Monte-Carlo runs without and with control (main.go:72-90)
    for tag, Fm, Gm in (("noctrl", F, G), ("ctrl", Fcl, Gcl)):
        mckf = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, x0, P0, Fm, Gm, H, Q, R, noise=k.NOISE_AWGN, seed=5044)   # ONE filter, main.go:74 / :85
        mc = ga.new_monte_carlo_runs(runs, SAMPLES, 2, zero_u, mckf)                                                  # main.go:76 / :86
        for i, contents in enumerate(mc.as_csv(headers)):                                                             # main.go:79-83 / :87-91
            with open(os.path.join(outdir, "mc-%s-%s.csv" % (tag, headers[i])), "w") as fh:
                fh.write(contents)
        if tag == "ctrl":
            truth_mc = mc
    # truth generation: one closed-loop pure predictor with AWGN (main.go:56-62, 92-101)
    truth = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, x0, P0, Fcl, Gcl, H, Q, R, flags=k.FLAG_FULL_ESTIMATE, noise=k.NOISE_AWGN, seed=7)
    state_truth, measurements = np.zeros((SAMPLES, 4)), np.zeros((SAMPLES, 2))
    texp = CSVExporter(headers, outdir, "truth.csv")
    for s in range(SAMPLES):
        est = truth.update(np.zeros(2), np.zeros(2))
        state_truth[s], measurements[s] = est.state()[0], est.measurement()[0]
        texp.write(state_truth[s], est.covariance()[0])
    texp.close()
    # the three filters on the truth's measurements; exported as error w.r.t. the truth (truth.go:16-40)
    filters = {
        "vanilla": ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, Fcl, Gcl, H, Q, R),
        "information": ga.FilterBatch.new_ldkf(k.INFORMATION, np.zeros(4), np.zeros((4, 4)), Fcl, Gcl, H, Q, R),
        "sqrt": ga.FilterBatch.new_ldkf(k.SQUAREROOT, x0, P0, Fcl, Gcl, H, Q, R),
    }
    exps = {n: CSVExporter(headers, outdir, n + ".csv") for n in filters}
    rms, history = {}, {}
    for name, kf in filters.items():
        err2, history[name] = np.zeros(4), np.zeros((SAMPLES, 4))
        for s in range(SAMPLES):
            est = kf.update(measurements[s], np.zeros(2))
            history[name][s] = est.state()[0]
            e = history[name][s] - state_truth[s]
            err2 += e * e
            exps[name].write(e, est.covariance()[0])
        exps[name].close()
        rms[name] = np.sqrt(err2 / SAMPLES)
    # chi-square on the closed-loop Monte-Carlo runs (main.go:163-175)
    chikf = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, Fcl, Gcl, H, Q, R)   # the reference hands over vanillaKF itself; NewChiSquare Reset()s it per run
    nis, nees = ga.new_chi_square(chikf, truth_mc, zero_u, True, True)      # main.go:165
    with open(os.path.join(outdir, "chisquare.csv"), "w") as fh:
        fh.write("NIS,NEES\n")
        for s in range(SAMPLES):
            fh.write("%f,%f\n" % (nis[s], nees[s]))
    return {"rms": rms, "history": history, "measurements": measurements, "nis_mean": float(nis.mean()), "nees_mean": float(nees.mean()), "mc_stddev_last": truth_mc.stddev(SAMPLES - 1), "mc": truth_mc}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=15)   # numMC := 15 (main.go:75)
    ap.add_argument("outdir", nargs="?", default="./statod5044_out")
    a = ap.parse_args()
    out = main(a.outdir, a.runs)
    print({kk: v for kk, v in out.items() if kk not in ("mc", "history", "measurements")})
