"""The reference's own LDKF smoke tests (vanilla_test.go:29-93, squareroot_test.go:29-91, information_test.go:45-119)
run against the engine: the 3-state "Midterm2" model (helper_test.go:17-22) with AWGN noise, 99 updates on the
reference's measurement sequence, setters, Reset, dimension errors.  The reference only logs 2-sigma breaches (its noise
is seeded by the wall clock); here the device's draws are replayed through the oracle, so every step is also checked."""
import os

import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
from oracle import oracle as orc
from tests.achieved import within

pytestmark = pytest.mark.gpu
YACC = np.loadtxt(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "midterm2_yacc.csv"), delimiter=",")

dt = 0.01
F = np.array([[1, 0.01, 5e-5], [0, 1, 0.01], [0, 0, 1]])
G = np.array([[(5e-7) / 3], [5e-5], [0.01]])
Q = np.array([[2.5e-15, 6.25e-13, (25e-11) / 3], [6.25e-13, (5e-7) / 3, 2.5e-8], [(25e-11) / 3, 2.5e-8, 5e-6]])
R = np.array([[0.005 / dt]])
H = np.array([[1.0, 0, 0]])
x0, P0 = np.array([0, 0.35, 0]), 10.0 * np.eye(3)


@pytest.mark.parametrize("kind,okind,tol", [(k.VANILLA, orc.VANILLA, 1e-9), (k.SQUAREROOT, orc.SQUAREROOT, 1e-9), (k.INFORMATION, orc.INFORMATION, 1e-7)])
def test_midterm2_awgn_run_setters_reset_and_dimension_errors(kind, okind, tol):
    flags = k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0
    kf = ga.FilterBatch.new_ldkf(kind, x0, P0, F, G, H, Q, R, noise=k.NOISE_AWGN, seed=2017, flags=flags)
    f = orc.Filter.information_from_state(x0, P0, F, G, H, Q, R) if kind == k.INFORMATION else orc.Filter.ldkf(okind, x0, P0, F, G, H, Q, R)
    # setters with the same values (vanilla_test.go:43-58)
    kf.set_state_transition(F); kf.set_input_control(G); kf.set_measurement_matrix(H); kf.set_noise(Q, R)
    f.set_state_transition(F); f.set_input_control(G); f.set_measurement_matrix(H); f.set_noise(Q, R)
    LQ, LR = orc.cholesky_lower(Q)[1], orc.cholesky_lower(R)[1]
    breaches = 0
    for step in range(1, 100):
        est = kf.update(np.array([YACC[step]]), np.zeros(1))
        t = step - 1
        w1, v, w2 = LQ @ kf.noise_sample(0, 0, t, 0, 3), LR @ kf.noise_sample(0, 0, t, 1, 1), LQ @ kf.noise_sample(0, 0, t, 2, 3)
        assert f.update(np.array([YACC[step]]), np.zeros(1), w1, v, w2) == orc.OK
        xs = est.state()[0]
        assert within(np.linalg.norm(xs - f.state()) / max(np.linalg.norm(f.state()), 1e-3), tol, "state"), step
        inside = bool(est.is_within_nsigma(2)[0])
        assert inside == bool(f.is_within_nsigma(2)), step
        breaches += (not inside)
    assert kf.step() == 99
    assert within(synth.rel_frobenius(est.covariance(), f.covariance()[None]), max(tol, 1e-8), "covariance")
    # Reset (vanilla_test.go:78-84)
    kf.reset()
    assert kf.step() == 0
    assert np.allclose(kf.get(k.STATE)[0], x0, atol=1e-12 if kind != k.INFORMATION else 1e-9)
    # invalid control / measurement vectors (vanilla_test.go:86-92)
    with pytest.raises(ga.KalmanError, match=r"dimensions must agree: control"):
        kf.update(np.zeros(1), np.zeros(2))
    with pytest.raises(ga.KalmanError, match=r"dimensions must agree: measurement"):
        kf.update(np.zeros(2), np.zeros(1))
