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


# ---- block-diagonal embedding ---------------------------------------------------------
# k copies of the 4-state system side by side: diag(F..), diag(H..), diag(Q..), diag(R..), x0 / P0 repeated, the control
# matrix STACKED (one control value drives every block, as in the 4-state run).  Every product of the three Updates
# (vanilla.go:138-205, squareroot.go:155-268, information.go:163-212) stays block-diagonal -- the Householder reflectors of
# squareroot.go:176-222 and the pivot choices of the LU inverses included -- so block b of the 4k-state result has to equal
# the 4-state run, i.e. the reference's own vanilla.csv / sqrt.csv / information.csv, whatever order a kernel evaluates the
# sums in.  4k = 8, 12, 16 land on the split-lane kernels (p = k on ordinary steps, 2k on every 10th).
def block_diag(M, k):
    M = np.atleast_2d(np.asarray(M, dtype=np.float64))
    r, c = M.shape
    out = np.zeros((k * r, k * c))
    for b in range(k):
        out[b * r:(b + 1) * r, b * c:(b + 1) * c] = M
    return out


def embedded(k):
    return dict(F=block_diag(F, k), G=np.tile(G, (k, 1)), H1=block_diag(H1, k), H2=block_diag(H2, k),
                Q=block_diag(Q, k), R1=block_diag(R1, k), R2=block_diag(R2, k),
                X0=np.tile(X0, k), P0=block_diag(P0, k))


# Information: `SetNoise` never refreshes the cached R^-1 (information.go:136-138), so the 4-state run multiplies H^T by the
# SCALAR 1/0.05 on every step, whatever H is (information.go:198-200).  A k x k cached R^-1 against a 2k-row H would panic in
# mat64 instead; the embedding therefore keeps 2k measurement rows throughout -- R = 0.05 * 1 (2k x 2k) at construction,
# H = diag(H1..) on every 10th step and diag(H1z..) otherwise, H1z = H1 with its position row zeroed and y = [0, yacc]: the same
# sums as the reference's with exact zeros added.
H1Z = np.array([[0.0, 0, 0, 0], [0, 0, 1, 1]])


def embedded_information(k):
    e = embedded(k)
    e.update(H1Z=block_diag(H1Z, k), RI=0.05 * np.eye(2 * k), X0=np.zeros(4 * k), P0=np.zeros((4 * k, 4 * k)))   # main.go:118-121
    return e


def run_protocol_embedded_information(k, update, set_h, row_of, steps=None):
    e = embedded_information(k)
    u, yacc, ypos = load_inputs()
    rows = [row_of()]
    K = len(yacc) if steps is None else steps
    for t in range(K):
        tenth = (t + 1) % 10 == 0
        if tenth:
            set_h(e["H1"])
        update(np.tile(np.array([ypos[t] if tenth else 0.0, yacc[t]]), k), np.array([u[t]]))
        rows.append(row_of())
        if tenth:
            set_h(e["H1Z"])
    return np.array(rows)


def run_protocol_embedded(k, update, set_h, set_noise, row_of, steps=None):
    """run_protocol for the k-block system: each block is fed the same measurements."""
    e = embedded(k)
    u, yacc, ypos = load_inputs()
    rows = [row_of()]
    K = len(yacc) if steps is None else steps
    for t in range(K):
        if (t + 1) % 10 == 0:
            set_h(e["H1"])
            set_noise(e["Q"], e["R1"])
            y = np.tile(np.array([ypos[t], yacc[t]]), k)
        else:
            y = np.tile(np.array([yacc[t]]), k)
        update(y, np.array([u[t]]))
        rows.append(row_of())
        if (t + 1) % 10 == 0:
            set_h(e["H2"])
            set_noise(e["Q"], e["R2"])
    return np.array(rows)


def export_rows_blocks(state, covar, k):
    """The exporter's row of every 4-state diagonal block: shape (k, 12)."""
    return np.array([export_row(state[4 * b:4 * b + 4], covar[4 * b:4 * b + 4, 4 * b:4 * b + 4]) for b in range(k)])


def off_block_max(M, k, r=4, c=4):
    """Largest magnitude outside the diagonal blocks (has to stay exactly zero)."""
    M = np.array(M, dtype=np.float64, copy=True)
    for b in range(k):
        M[b * r:(b + 1) * r, b * c:(b + 1) * c] = 0.0
    return float(np.max(np.abs(M))) if M.size else 0.0
