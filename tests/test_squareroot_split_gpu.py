"""SquareRoot.Update (squareroot.go:129-274) beyond 6 states: the kernel that splits ONE filter over four lanes with the Householder
panels distributed by columns (gokalman_amd/csrc/kb_squareroot_split.h) against the CPU oracle, through the C ABI: 12 / 6 at 4096
filters x 20 steps (<= 1e-9 relative Frobenius on x and on the covariance S S^T), the padded family 7..16 states (eight lanes per filter beyond 12) with / without
KB_FLAG_FULL_ESTIMATE, control input, AWGN replayed through the oracle, and equality with the statement kernel to 1e-12."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth
from oracle import oracle as orc
from tests.test_vanilla_split_gpu import _model
from tests.achieved import within

pytestmark = pytest.mark.gpu
TOL = 1e-9


def test_sqsplit_12x6_at_4096_filters_20_steps_vs_oracle():
    import torch
    N, steps = 4096 + 21, 20
    d = synth.linear_batch(N, 12, 6, steps)
    b = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    for t in range(steps):
        b.update_dev(y[t].data_ptr(), N)
    b.synchronize()
    assert b.step() == steps and not b.status().any()
    xo, Po, nerr = orc.ldkf_batch(orc.SQUAREROOT, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
    assert nerr == 0
    assert synth.rel_frobenius(b.get(k.STATE), xo) <= TOL
    assert synth.rel_frobenius(b.get(k.COVAR), Po) <= TOL
    s = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"][:256], d["P0"][:256], d["F"][:256], None, d["H"][:256], d["Q"][:256], d["R"][:256], flags=k.FLAG_STATEMENT_KERNELS)
    for t in range(steps):
        s.update(d["y"][t, :256])
    assert synth.rel_frobenius(b.get(k.RAW_MAT, 0, 256), s.get(k.RAW_MAT)) <= 1e-11   # the factor itself, not only S S^T
    assert synth.rel_frobenius(b.get(k.STATE, 0, 256), s.get(k.STATE)) <= 1e-11


@pytest.mark.parametrize("n,p,m", [(7, 2, 0), (8, 4, 1), (9, 5, 2), (10, 1, 0), (11, 7, 0), (12, 6, 0), (12, 8, 2), (12, 3, 1),
                                   (13, 2, 0), (14, 8, 1), (15, 7, 2), (16, 8, 2), (16, 6, 0), (8, 3, 2), (7, 4, 0), (8, 5, 0), (10, 4, 0), (16, 4, 1), (14, 6, 0)])
@pytest.mark.parametrize("full", [False, True])
def test_sqsplit_padded_family_vs_oracle(n, p, m, full):
    N, steps = 150, 6
    d = _model(N, n, p, m, steps, 7000 + 100 * n + 10 * p + m)
    b = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE if full else 0)
    fs = [orc.Filter.ldkf(orc.SQUAREROOT, d["x0"][i], d["P0"][i], d["F"][i], None if m == 0 else d["G"][i], d["H"][i], d["Q"][i], d["R"][i]) for i in range(N)]
    for t in range(steps):
        est = b.update(d["y"][t], None if m == 0 else d["u"][t])
        for i, f in enumerate(fs):
            assert f.update(d["y"][t, i], None if m == 0 else d["u"][t, i]) == orc.OK
    assert not b.status().any()
    assert synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs])) <= TOL
    assert synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs])) <= TOL
    if full:
        assert synth.rel_frobenius(est.pred_covariance(), np.array([f.pred_covariance() for f in fs])) <= TOL
        assert synth.rel_frobenius(est.gain(), np.array([f.gain() for f in fs])) <= TOL
        assert within(np.max(np.abs(est.innovation() - np.array([f.innovation() for f in fs]))), 1e-8)
        assert within(np.max(np.abs(est.measurement() - np.array([f.measurement() for f in fs]))), 1e-8)


@pytest.mark.parametrize("n,p,m,full", [(12, 6, 0, False), (12, 6, 0, True), (9, 3, 1, True), (11, 8, 2, False), (16, 8, 2, True), (14, 5, 0, False),
                                        (8, 4, 0, False), (7, 2, 1, True), (16, 4, 0, False), (10, 4, 2, True), (15, 6, 1, True)])
def test_sqsplit_awgn_replayed_through_the_oracle(n, p, m, full):
    """AWGN: Measurement(k) into yhat (squareroot.go:239), Process(k) into x+ (:268); the device's draws replayed through the oracle."""
    N, steps = 150, 5
    d = _model(N, n, p, m, steps, 555 + n + p)
    b = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE if full else 0,
                                noise=k.NOISE_AWGN, seed=77)
    for t in range(steps):
        est = b.update(d["y"][t], d["u"][t] if m else None, snapshot=(t == steps - 1))
    xs, Ps, ys = [], [], []
    for i in range(N):
        LQ, LR = orc.cholesky_lower(d["Q"][i])[1], orc.cholesky_lower(d["R"][i])[1]
        f = orc.Filter.ldkf(orc.SQUAREROOT, d["x0"][i], d["P0"][i], d["F"][i], d["G"][i] if m else None, d["H"][i], d["Q"][i], d["R"][i])
        for t in range(steps):
            v, w2 = LR @ b.noise_sample(i, 0, t, 1, p), LQ @ b.noise_sample(i, 0, t, 2, n)
            assert f.update(d["y"][t, i], d["u"][t, i] if m else None, v_meas=v, w_post=w2) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance()); ys.append(f.measurement())
    assert synth.rel_frobenius(est.state(), np.array(xs)) <= TOL
    assert synth.rel_frobenius(est.covariance(), np.array(Ps)) <= TOL
    if full:
        assert synth.rel_frobenius(est.measurement(), np.array(ys)) <= TOL
    assert not b.status().any() and b.step() == steps
