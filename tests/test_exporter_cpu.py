"""gokalman_amd.exporter.CSVExporter writes the reference's CSV format (exporter.go:34-91, TestCSVExport): driven here
by the CPU oracle on the jerkcar scenario, the file must equal the reference's committed vanilla.csv line for line
(but for the date comments and last-digit %f rounding ties)."""
import os

import numpy as np

from gokalman_amd.exporter import CSVExporter
from oracle import oracle as orc
from tests import jerkcar as jc


def test_csv_exporter_reproduces_reference_file(tmp_path):
    u, yacc, ypos = jc.load_inputs()
    f = orc.Filter.ldkf(orc.VANILLA, jc.X0, jc.P0, jc.F, jc.G, jc.H2, jc.Q, jc.R2)
    ex = CSVExporter(["position", "velocity", "acceleration", "bias"], str(tmp_path), "vanilla.csv")
    ex.write(f.state(), f.covariance())
    for k in range(len(yacc)):
        if (k + 1) % 10 == 0:
            f.set_measurement_matrix(jc.H1); f.set_noise(jc.Q, jc.R1)
            assert f.update(np.array([ypos[k], yacc[k]]), np.array([u[k]])) == orc.OK
        else:
            assert f.update(np.array([yacc[k]]), np.array([u[k]])) == orc.OK
        ex.write(f.state(), f.covariance())
        if (k + 1) % 10 == 0:
            f.set_measurement_matrix(jc.H2); f.set_noise(jc.Q, jc.R2)
    ex.close()
    ours = [l.rstrip("\n") for l in open(ex.path)]
    ref = [l.rstrip("\n") for l in open(os.path.join(jc.GOLDEN, "vanilla.csv"))]
    assert ours[0].startswith("# Creation date (UTC): ") and ref[0].startswith("# Creation date (UTC): ")
    assert ours[1] == ref[1]                                   # header with the +2s / -2s columns
    body_o = [l for l in ours[2:] if l and not l.startswith("#")]
    body_r = [l for l in ref[2:] if l and not l.startswith("#")]
    assert len(body_o) == len(body_r) == 2001
    same = sum(a == b for a, b in zip(body_o, body_r))
    assert same >= 0.98 * len(body_r), same
    assert any(l.startswith("# Closing date (UTC): ") for l in ours)


def test_underscore_headers_have_no_covariance_columns(tmp_path):
    ex = CSVExporter(["x", "_raw"], str(tmp_path), "t.csv", covar_bound=3)
    ex.close()
    assert open(ex.path).readlines()[1].strip() == "x,x+3s,x-3s,raw"
