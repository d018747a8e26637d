"""String() of the reference's filters, noises and estimates (vanilla.go:76-78, :276-284; information.go:96-98, :318-325;
squareroot.go:65-67, :347-355; srif.go:283-289; hybrid.go:63-65, :300-308; noise.go:62-64, :104-106, :162-164).

The reference prints matrices through gonum's `mat64.Formatted(m, mat64.Prefix(p))` with the `%v` verb: box-drawing brackets
(square brackets for a single row), every element right-aligned to the widest one, two spaces between columns, the prefix
in front of every line but the first.  That layout is restated here from gonum's documented behaviour; gonum itself is not
available in this environment, so byte-for-byte equality with a Go run is unverified (the field order, labels and prefixes
are the reference's own format strings).
"""
import math

import numpy as np


def go_v(x):
    """fmt's %v of a float64 = strconv's 'g' with the shortest digits that round-trip: %e form for decimal exponents < -4 or
    >= 6 (ftoa.go: "if precision was the shortest possible, use precision 6 for this decision"), so fmt.Println(1e6) prints
    1e+06 and 123456789.0 prints 1.23456789e+08.  (The threshold 21 belongs to encoding/json, not to fmt.)"""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "+Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0" if math.copysign(1.0, x) < 0 else "0"
    r = repr(abs(x))
    if "e" in r:
        mant, ex = r.split("e")
        ex = int(ex)
    else:
        mant, ex = r, 0
    if "." in mant:
        ip, fp = mant.split(".")
    else:
        ip, fp = mant, ""
    if fp == "0":
        fp = ""
    digits = (ip + fp).lstrip("0")
    # decimal exponent of the first significant digit
    lead = len(ip.lstrip("0")) if ip.strip("0") else -(len(fp) - len(fp.lstrip("0")))
    e10 = ex + (lead - 1 if ip.strip("0") else lead - 1)
    digits = digits.rstrip("0") or "0"
    sign = "-" if x < 0 else ""
    if e10 < -4 or e10 >= 6:
        m = digits[0] + ("." + digits[1:] if len(digits) > 1 else "")
        return "%s%se%s%02d" % (sign, m, "-" if e10 < 0 else "+", abs(e10))
    if e10 >= 0:
        if len(digits) <= e10 + 1:
            return sign + digits + "0" * (e10 + 1 - len(digits))
        return sign + digits[: e10 + 1] + "." + digits[e10 + 1:]
    return sign + "0." + "0" * (-e10 - 1) + digits


def formatted(m, prefix=""):
    """mat64.Formatted(m, mat64.Prefix(prefix)) with %v; vectors are n x 1 matrices, None prints as Go's nil matrix value."""
    if m is None:
        return "<nil>"
    a = np.asarray(m, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    rows, cols = a.shape
    cells = [[go_v(a[i, j]) for j in range(cols)] for i in range(rows)]
    width = max(len(c) for r in cells for c in r) if rows and cols else 0
    lines = []
    for i, r in enumerate(cells):
        body = "  ".join(c.rjust(width) for c in r)
        if rows == 1:
            left, right = "[", "]"
        elif i == 0:
            left, right = "⎡", "⎤"
        elif i == rows - 1:
            left, right = "⎣", "⎦"
        else:
            left, right = "⎢", "⎥"
        lines.append(left + body + right)
    return ("\n" + prefix).join(lines)


def estimate_string(kind_name, state, meas, covar, gain, pred_covar, innov):
    """<Kind>Estimate.String().  kind_name in {vanilla, squareroot, hybrid, information, srif}."""
    s, y, P = formatted(state, "  "), formatted(meas, "  "), formatted(covar, "  ")
    Pm = formatted(pred_covar, "   ")
    if kind_name in ("vanilla", "squareroot", "hybrid"):
        return "{\ns=%s\ny=%s\nP=%s\nK=%s\nP-=%s\ni=%s\n}" % (s, y, P, formatted(gain, "  "), Pm, formatted(innov, "  "))
    if kind_name == "information":
        return "{\ns=%s\ny=%s\nP=%s\nP-=%s\ni=%s\n}" % (s, y, P, Pm, formatted(innov, "  "))
    if kind_name == "srif":
        return "{\ns=%s\ny=%s\nP=%s\nP-=%s\n}" % (s, y, P, Pm)
    raise ValueError(kind_name)


def noise_string(noise_kind, Q, R):
    """Noiseless / AWGN / BatchNoise String() (noise.go)."""
    if noise_kind == "batch":
        return "BatchNoise"
    name = "AWGN" if noise_kind == "awgn" else "Noiseless"
    return "%s{\nQ=%s\nR=%s}\n" % (name, formatted(Q, "  "), formatted(R, "  "))


def filter_string(kind_name, F, G, H, noise, step=0):
    """Vanilla / SquareRoot / Information / HybridKF String().  For Information pass F^-1 as F."""
    if kind_name == "hybrid":
        return "HybridKF [k=%d]\n%s" % (step, noise)
    if kind_name == "information":
        return "inv(F)=%s\nG=%s\nH=%s\n%s" % (formatted(F, "      "), formatted(G, "  "), formatted(H, "  "), noise)
    return "F=%s\nG=%s\nH=%s\n%s" % (formatted(F, "  "), formatted(G, "  "), formatted(H, "  "), noise)
