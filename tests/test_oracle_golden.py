"""Pins the CPU oracle against every golden vector / KAT the reference holds for the hot path."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import jerkcar as jc

PRINT_TOL = 5.1e-7   # the CSVs are printed with %f (6 decimals)


def _replay(filt):
    def row():
        return jc.export_row(filt.state(), filt.covariance())

    def upd(y, u):
        rc = filt.update(y, u)
        assert rc == orc.OK, rc

    return jc.run_protocol(upd, filt.set_measurement_matrix, filt.set_noise, row)


def test_jerkcar_vanilla():
    f = orc.Filter.ldkf(orc.VANILLA, jc.X0, jc.P0, jc.F, jc.G, jc.H2, jc.Q, jc.R2)
    got, exp = _replay(f), jc.load_expected("vanilla")
    assert got.shape == exp.shape == (2001, 12)
    assert np.max(np.abs(got - exp)) <= PRINT_TOL


def test_jerkcar_squareroot():
    f = orc.Filter.ldkf(orc.SQUAREROOT, jc.X0, jc.P0, jc.F, jc.G, jc.H2, jc.Q, jc.R2)
    got, exp = _replay(f), jc.load_expected("sqrt")
    assert got.shape == exp.shape
    assert np.max(np.abs(got - exp)) <= PRINT_TOL


def test_jerkcar_information():
    f = orc.Filter.ldkf(orc.INFORMATION, np.zeros(4), np.zeros((4, 4)), jc.F, jc.G, jc.H2, jc.Q, jc.R2)
    got, exp = _replay(f), jc.load_expected("information")
    assert got.shape == exp.shape
    assert np.max(np.abs(got - exp)) <= PRINT_TOL


def test_householder_kat():
    # helper_test.go:108-117
    A = np.array([[1.0, -2, -1], [2, -1, 1], [1, 1, 2]])
    exp = np.array([[-2.449489742783178, 1.224744871391589, -1.2247448713915892],
                    [0, -2.121320343559643, -2.121320343559643], [0, 0, 0]])
    got = orc.householder_transf(A, 2, 1)
    assert np.max(np.abs(got - exp)) <= 1e-15


def test_srif_measurement_update_kat():
    # srif_test.go:31-56
    R = np.array([[0.1, 0], [0, 0.1]])
    H = np.array([[1.0, -2], [2, -1], [1, 1]])
    Rk, bk, ek = orc.measurement_srif_update(R, H, [0.2, 0.2], [-1.1, 1.2, 1.8])
    assert np.max(np.abs(ek - [-0.1319, 0.0871, -0.2810])) <= 1e-4
    assert np.max(np.abs(bk - [-1.2727, -2.0607])) <= 1e-4
    assert np.max(np.abs(Rk - [[-2.4515, 1.2237], [0, -2.1243]])) <= 1e-4


def test_srif_r0_roundtrip():
    # srif_test.go:15-29
    x0 = np.array([0, 0.35, 0])
    P0 = 10.0 * np.eye(3)
    R = np.diag([5e-3 ** 2, 5e-6 ** 2])
    f = orc.Filter.srif(x0, P0, R, 2, non_tri_r=True)  # measSize (3 in the test) only sizes Predict() zeros
    assert np.max(np.abs(f.covariance() - P0)) <= 1e-12


def test_as_sym_dense():
    # helper_test.go:66-92
    rc, s = orc.as_sym_dense(np.array([[1, 0.1, 2], [0.1, 3, 5], [2, 5, 7.0]]))
    assert rc == orc.OK and np.array_equal(s, s.T)
    rc, _ = orc.as_sym_dense(np.array([[1.0, 0, 3], [0, 1, 0], [1, 2, 1]]))
    assert rc == orc.ERR_ASYMMETRIC
    # upper triangle wins, no averaging
    rc, s = orc.as_sym_dense(np.array([[1.0, 2.0], [2.0 + 1e-7, 1.0]]))
    assert rc == orc.OK and s[1, 0] == 2.0


def test_sign_deadband():
    assert orc.sign(1e-13) == 1.0 and orc.sign(-1e-13) == 1.0
    assert orc.sign(-2.0) == -1.0 and orc.sign(3.0) == 1.0


def test_quirk_sensitivity_squareroot():
    """Without the Uc quirk + LAPACK signs the fixture is missed by >1e-3: the pin is sharp."""
    exp = jc.load_expected("sqrt")
    f = orc.Filter.ldkf(orc.SQUAREROOT, jc.X0, jc.P0, jc.F, jc.G, jc.H2, jc.Q, jc.R2)
    got = _replay(f)
    # position +2sigma column moves a lot over the run; a textbook SRKF differs there
    assert np.max(np.abs(exp[:, 1] - exp[0, 1])) > 1e-3
    assert np.max(np.abs(got[:, 1] - exp[:, 1])) <= PRINT_TOL
