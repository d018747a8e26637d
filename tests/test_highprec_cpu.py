"""The arbiter above fp64 (tests/golden/make_highprec.py -> tests/golden/generated/*_hp.npz, 60-digit restatements of hybrid.go:104-204,
vanilla.go:128-220 + noise.go:67-106, squareroot.go:129-274) against the ORACLE: what pins the restatements (a well-conditioned control
on which both must agree to rounding) and what the oracle's own error against the exact result is on the ill-conditioned legs -- the
yardstick tests/test_highprec_gpu.py and bench.py hold the engine to (engine error <= 4 x oracle error)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import highprec as hp
from tests.achieved import within


def test_restatements_are_pinned_by_the_oracle_on_a_well_conditioned_control():
    z = hp.load("ldkf_wellcond_6x3")
    for kind, name, tol in ((orc.VANILLA, "vanilla", 1e-13), (orc.SQUAREROOT, "squareroot", 5e-12)):
        xs, Ps = hp.oracle_ldkf(orc, kind, z)
        for t in range(xs.shape[0]):
            assert hp.rel_err(xs[t], z["x_" + name][t]).max() <= tol, (name, t)
            assert hp.rel_err(Ps[t], z["P_" + name][t]).max() <= tol, (name, t)   # (a wrong Householder sign convention would show as O(1) here)


@pytest.mark.parametrize("name", ["hybrid_ekf_bench_6x2", "hybrid_ckf_bench_6x2", "hybrid_ekf_stm_6x2"])
def test_oracle_error_against_the_exact_result_on_config_d_ii(name):
    """configs[3] D(ii) as SURVEY 8d specifies it (R = diag(1e-6), P0 = diag(10, 10, 10, 1, 1, 1)): the reference-order fp64 evaluation
    itself is 1e-5 away from the exact result on the worst of 64 filters (median 1e-9) -- two fp64 evaluations can differ by that much
    and both be right.  Step 1 is exact to rounding on every filter."""
    z = hp.load(name)
    xs, Ps = hp.oracle_hybrid(orc, z)
    assert hp.rel_err(xs[0], z["x"][0]).max() <= 1e-13 and hp.rel_err(Ps[0], z["P"][0]).max() <= 1e-13
    ex = np.array([hp.rel_err(xs[t], z["x"][t]) for t in range(xs.shape[0])])
    eP = np.array([hp.rel_err(Ps[t], z["P"][t]) for t in range(xs.shape[0])])
    print("%s: oracle vs exact over 20 steps: x max %.2e median %.2e; P max %.2e median %.2e" % (name, ex.max(), np.median(ex), eP.max(), np.median(eP)))
    assert within(float(ex.max()), 1e-3) and within(float(eP.max()), 1e-3)
    assert np.median(ex) <= 1e-8 and np.median(eP) <= 1e-8


@pytest.mark.parametrize("name", ["vanilla_batchnoise_6x3", "vanilla_batchnoise_12x6"])
def test_batch_noise_has_an_exact_answer_for_n_measurements_and_none_after(name):
    """BatchNoise reports zero Q and R (noise.go:89-98).  FINDING of the arbiter: after n / p steps the exact covariance is the ZERO matrix
    (|P| ~ 1e-61 at 60 digits), and from the next step on H P- H^T + R is exactly singular -- in exact arithmetic the reference's Update
    returns its "could not invert" error there (vanilla.go:164-167).  No fp64 evaluation sees that (P is 1e-16 of rounding noise and
    the step amplifies it): past n measurements there is NO exact result to be close to, for the oracle or for the engine; up to
    there the state is good to 1e-14 and P to 1e-16 |P0|."""
    z = hp.load(name)
    n, p = z["x0"].shape[1], z["y"].shape[2]
    exact_steps = n // p
    assert z["valid"][:exact_steps].all() and not z["valid"][exact_steps:].any()
    assert np.abs(z["P"][exact_steps - 1]).max() <= 1e-50
    xs, Ps, rcs = hp.oracle_batchnoise(orc, z)
    p0 = np.linalg.norm(z["P0"].reshape(len(z["P0"]), -1), axis=1)
    for t in range(exact_steps):
        assert (rcs[t] == orc.OK).all()
        assert hp.rel_err(xs[t], z["x"][t]).max() <= 1e-13
        assert hp.rel_err(Ps[t], z["P"][t], p0).max() <= 1e-15
    print("%s: oracle return codes past the last exact step: %s" % (name, sorted(set(rcs[exact_steps:].ravel().tolist()))))


def test_oracle_error_on_the_ill_conditioned_linear_twin():
    z = hp.load("ldkf_illcond_6x3")
    for kind, name, tol in ((orc.VANILLA, "vanilla", 1e-11), (orc.SQUAREROOT, "squareroot", 1e-10)):
        xs, Ps = hp.oracle_ldkf(orc, kind, z)
        ex = max(hp.rel_err(xs[t], z["x_" + name][t]).max() for t in range(xs.shape[0]))
        eP = max(hp.rel_err(Ps[t], z["P_" + name][t]).max() for t in range(xs.shape[0]))
        print("ldkf_illcond %s: oracle vs exact x %.2e P %.2e" % (name, ex, eP))
        assert ex <= tol and eP <= tol


@pytest.mark.parametrize("name", ["srif_12x6", "srif_7x3"])
def test_two_independent_readings_of_srif_go_agree(name):
    """srif.go:101-160 has no reference fixture that runs without the external `smd` package (srif_test.go:11): the oracle's C restatement
    and the 60-digit Python restatement of tests/golden/make_highprec.py were written independently from the Go source (time update,
    the chol_L(R) quirk of :47, whitening, HouseholderTransf with Sign()'s dead band) -- on config E's generator they agree to rounding."""
    z = hp.load(name)
    T, N = z["Phi"].shape[:2]
    p = z["real"].shape[2]
    eb = eR = 0.0
    for i in range(N):
        f = orc.Filter.srif(z["x0"][i], z["P0"][i], z["R"][i], p)
        for t in range(T):
            f.prepare(z["Phi"][t, i], z["Ht"][t, i])
            assert f.update_nl(z["real"][t, i], z["comp"][t, i]) == orc.OK
            eb = max(eb, float(np.linalg.norm(f.raw_vec() - z["b"][t, i]) / np.linalg.norm(z["b"][t, i])))
            eR = max(eR, float(np.linalg.norm(f.raw_mat() - z["Rk"][t, i]) / np.linalg.norm(z["Rk"][t, i])))
    print("%s: oracle against the 60-digit restatement: b %.2e R %.2e" % (name, eb, eR))
    assert eb <= 1e-13 and eR <= 1e-13
