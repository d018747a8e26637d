#!/usr/bin/env python3
"""examples/jerkcar of the reference (examples/jerkcar/main.go) on the MI355X engine: three filters in
lock-step (Vanilla, Information, SquareRoot), H and the noise swapped on every 10th step, outputs written
with the reference's CSV exporter format.  usage: python examples/jerkcar.py [outdir]
The inputs are the reference's own data files (tests/golden/jerkcar/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd.exporter import CSVExporter
from tests import jerkcar as jc


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    u, yacc, ypos = jc.load_inputs()
    headers = ["position", "velocity", "acceleration", "bias"]
    filters = {
        "vanilla": ga.FilterBatch.new_ldkf(k.VANILLA, jc.X0, jc.P0, jc.F, jc.G, jc.H2, jc.Q, jc.R2, pmax=2),
        "information": ga.FilterBatch.new_ldkf(k.INFORMATION, np.zeros(4), np.zeros((4, 4)), jc.F, jc.G, jc.H2, jc.Q, jc.R2, pmax=2),
        "sqrt": ga.FilterBatch.new_ldkf(k.SQUAREROOT, jc.X0, jc.P0, jc.F, jc.G, jc.H2, jc.Q, jc.R2, pmax=2),
    }
    exporters = {name: CSVExporter(headers, outdir, name + ".csv") for name in filters}
    for name, kf in filters.items():
        exporters[name].write(kf.get(k.STATE)[0], kf.get(k.COVAR)[0])
    for kk in range(len(yacc)):
        for name, kf in filters.items():
            if (kk + 1) % 10 == 0:                      # main.go:141-147
                kf.set_measurement_matrix(jc.H1)
                kf.set_noise(jc.Q, jc.R1)
                meas = np.array([ypos[kk], yacc[kk]])
            else:
                meas = np.array([yacc[kk]])
            est = kf.update(meas, np.array([u[kk]]))
            exporters[name].write(est.state()[0], est.covariance()[0])
            if (kk + 1) % 10 == 0:                      # main.go:155-159
                kf.set_measurement_matrix(jc.H2)
                kf.set_noise(jc.Q, jc.R2)
    for e in exporters.values():
        e.close()
    return {name: e.path for name, e in exporters.items()}


if __name__ == "__main__":
    print(main(sys.argv[1] if len(sys.argv) > 1 else "./jerkcar_out"))
