"""The reference's examples/jerkcar scenario as data + protocol.

Inputs and expected outputs are the reference's own committed files
(tests/golden/jerkcar/*.csv, copied byte-for-byte; md5 in tests/golden/README.md).
Model constants and the H/noise switching protocol follow
examples/jerkcar/main.go:94-161.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jerkcar")

F = np.array([[1, 0.01, 0.00005, 0], [0, 1, 0.01, 0], [0, 0, 1, 0], [0, 0, 0, 1.0005125020836]])
G = np.array([[0.0], [0.0001], [0.01], [0.0]])
H1 = np.array([[1.0, 0, 0, 0], [0, 0, 1, 1]])
H2 = np.array([[0.0, 0, 1, 1]])
Q = 1e-3 * np.array([
    [0.0000000000025, 0.000000000625, 0.000000083333333, 0],
    [0.000000000625, 0.000000166666667, 0.000025, 0],
    [0.000000083333333, 0.000025, 0.005, 0],
    [0, 0, 0, 0.530265088355421]])
R1 = np.array([[0.5, 0], [0, 0.05]])
R2 = np.array([[0.05]])
X0 = np.array([0, 0.45, 0, 0.09])
P0 = 10.0 * np.eye(4)


def load_inputs():
    u = np.loadtxt(os.path.join(GOLDEN, "uvec.csv"))          # one value per line
    with open(os.path.join(GOLDEN, "yacchist.csv")) as fh:
        yacc = np.array([float(v) for v in fh.readline().strip().split(",")])
    with open(os.path.join(GOLDEN, "yposhist.csv")) as fh:
        ypos = np.array([float(v) for v in fh.readline().strip().split(",")])
    ypos = np.where(np.isnan(ypos), 0.0, ypos)                # main.go:58-60
    return u, yacc, ypos


def load_expected(name):
    """2 header lines, then rows of (x_i, +2sigma_i, -2sigma_i) per component (exporter.go:34-45)."""
    rows = []
    with open(os.path.join(GOLDEN, name + ".csv")) as fh:
        for line in fh.readlines()[2:]:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            rows.append([float(v) for v in line.split(",")])
    return np.array(rows)


def export_row(state, covar):
    """CSVExporter.Write with covarBound = 2."""
    out = []
    for i in range(len(state)):
        b = 2.0 * np.sqrt(covar[i, i])
        out += [state[i], b, -b]
    return out


def run_protocol(update, set_h, set_noise, row_of, steps=None):
    """main.go:136-161: every 10th step swap in H1/noise1 with a 2-vector measurement."""
    u, yacc, ypos = load_inputs()
    rows = [row_of()]
    K = len(yacc) if steps is None else steps
    for k in range(K):
        if (k + 1) % 10 == 0:
            set_h(H1)
            set_noise(Q, R1)
            y = np.array([ypos[k], yacc[k]])
        else:
            y = np.array([yacc[k]])
        update(y, np.array([u[k]]))
        rows.append(row_of())
        if (k + 1) % 10 == 0:
            set_h(H2)
            set_noise(Q, R2)
    return np.array(rows)
