"""Committed oracle-generated vectors (tests/golden/make_generated.py): the oracle must still
reproduce them on CPU, and the HIP path must match them on the GPU."""
import os

import numpy as np
import pytest

from gokalman_amd import synth
from oracle import oracle as orc
from tests.achieved import within

GEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generated")
LDKF = [("vanilla_6x3", orc.VANILLA), ("squareroot_6x3", orc.SQUAREROOT), ("information_6x3", orc.INFORMATION)]
NL = [("srif_12x6", orc.SRIF, False), ("hybrid_ckf_6x2", orc.HYBRID, False), ("hybrid_ekf_6x2", orc.HYBRID, True)]


@pytest.mark.parametrize("name,kind", LDKF)
def test_oracle_reproduces_ldkf_vectors(name, kind):
    d = np.load(os.path.join(GEN, name + ".npz"))
    for i in range(0, 64, 16):
        if kind == orc.INFORMATION:
            f = orc.Filter.information_from_state(d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
        else:
            f = orc.Filter.ldkf(kind, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
        for t in range(50):
            f.update(d["y"][t, i])
            if t in (0, 9, 49):
                k = [0, 9, 49].index(t)
                assert np.allclose(f.state(), d["x_steps"][k, i], rtol=1e-12, atol=1e-14)
                assert np.allclose(f.covariance(), d["P_steps"][k, i], rtol=1e-11, atol=1e-16)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind", LDKF)
def test_gpu_matches_ldkf_vectors(name, kind):
    import gokalman_amd as ga
    from gokalman_amd import _capi as k
    d = np.load(os.path.join(GEN, name + ".npz"))
    gk = {orc.VANILLA: k.VANILLA, orc.SQUAREROOT: k.SQUAREROOT, orc.INFORMATION: k.INFORMATION}[kind]
    flags = k.FLAG_INFO_FROM_STATE if kind == orc.INFORMATION else 0
    b = ga.FilterBatch.new_ldkf(gk, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=flags)
    tol = 1e-7 if kind == orc.INFORMATION else 1e-9   # information form: covariance comes out of an inverse
    for t in range(50):
        b.update(d["y"][t])
        if t in (0, 9, 49):
            kk = [0, 9, 49].index(t)
            assert within(synth.rel_frobenius(b.get(k.STATE), d["x_steps"][kk]), tol, "state")
            assert within(synth.rel_frobenius(b.get(k.COVAR), d["P_steps"][kk]), tol, "covariance")


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,ekf", NL)
def test_gpu_matches_nldkf_vectors(name, kind, ekf):
    import gokalman_amd as ga
    from gokalman_amd import _capi as k
    d = np.load(os.path.join(GEN, name + ".npz"))
    N, n = d["x0"].shape
    p = d["R"].shape[-1]
    b = ga.FilterBatch(k.SRIF if kind == orc.SRIF else k.HYBRID, n, p, 0, N)
    b.set(k.X, d["x0"], 1); b.set(k.P, d["P0"], 2); b.set(k.R, d["R"], 2, p_rows=p); b.init()
    if ekf:
        b.enable_ekf()
    for t in range(d["Phi"].shape[0]):
        b.prepare(d["Phi"][t], d["Ht"][t])
        b.update_nl(d["real"][t], d["comp"][t])
    assert within(synth.rel_frobenius(b.get(k.STATE), d["x_final"]), 1e-8)
    assert within(synth.rel_frobenius(b.get(k.COVAR), d["P_final"]), 1e-8)
