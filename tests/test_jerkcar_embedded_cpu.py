"""The oracle on the block-diagonal embedding of the reference's jerkcar scenario (k = 2, 3, 4 copies: 8, 12, 16 states)
against the reference's own vanilla.csv / sqrt.csv / information.csv, block by block (tests/jerkcar.py, `embedded`).
This is the CPU half of the pin that reaches the split-lane kernels (tests/test_jerkcar_embedded_gpu.py is the other)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import jerkcar as jc

PRINT_TOL = 5.1e-7   # the CSVs are printed with %f (6 decimals)
KINDS = [("vanilla", orc.VANILLA), ("sqrt", orc.SQUAREROOT), ("information", orc.INFORMATION)]


@pytest.mark.parametrize("k", [2, 3, 4])
@pytest.mark.parametrize("name,kind", KINDS)
def test_oracle_on_embedded_jerkcar_equals_reference_csv(name, kind, k):
    info = kind == orc.INFORMATION
    e = jc.embedded_information(k) if info else jc.embedded(k)
    f = orc.Filter.ldkf(kind, e["X0"], e["P0"], e["F"], e["G"], e["H1Z"] if info else e["H2"], e["Q"], e["RI"] if info else e["R2"])
    off = [0.0]

    def row():
        P = f.covariance()
        off[0] = max(off[0], jc.off_block_max(P, k))
        return jc.export_rows_blocks(f.state(), P, k)

    def upd(y, u):
        assert f.update(y, u) == orc.OK

    if info:
        got = jc.run_protocol_embedded_information(k, upd, f.set_measurement_matrix, row)
    else:
        got = jc.run_protocol_embedded(k, upd, f.set_measurement_matrix, f.set_noise, row)   # (2001, k, 12)
    exp = jc.load_expected(name)
    assert got.shape == (2001, k, 12)
    err = max(np.max(np.abs(got[:, b] - exp)) for b in range(k))
    # the blocks do not interact: identical digits in every block, exact zeros between them
    spread = max(np.max(np.abs(got[:, b] - got[:, 0])) for b in range(k))
    print("embedded x%d %s: max |oracle - csv| %.3e, block spread %.3e, off-block %.3e" % (k, name, err, spread, off[0]))
    assert err <= PRINT_TOL
    assert off[0] == 0.0
    assert spread <= 1e-12
