"""Information.Update (information.go:153-227) beyond 6 states: the kernel that splits ONE filter over four / eight lanes
(gokalman_amd/csrc/kb_information_split.h; state-only outputs, the pivoted LU solve distributed over the lanes) against the CPU
oracle, through the C ABI: 12 / 6 at 4096 filters x 20 steps, the padded family 7..16 states with control input, pivoting forced
by a model whose M + Q^-1 has small diagonal entries, and equality with the statement kernel to rounding."""
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


def _oracle(d, steps, m):
    N = d["x0"].shape[0]
    fs = [orc.Filter.information_from_state(d["x0"][i], d["P0"][i], d["F"][i], None if m == 0 else d["G"][i], d["H"][i], d["Q"][i], d["R"][i]) for i in range(N)]
    for t in range(steps):
        for i, f in enumerate(fs):
            assert f.update(d["y"][t, i], None if m == 0 else d["u"][t, i]) == orc.OK
    return fs


def test_infsplit_12x6_at_4096_filters_20_steps_vs_oracle():
    import torch
    N, steps = 4096 + 21, 20
    d = synth.linear_batch(N, 12, 6, steps)
    b = ga.FilterBatch.new_ldkf(k.INFORMATION, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_INFO_FROM_STATE)
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    for t in range(steps):
        b.update_dev(y[t].data_ptr(), N)
    b.synchronize()
    assert b.step() == steps and not b.status().any()
    d["G"] = d["u"] = None
    sub = {kk: (v[:300] if kk != "y" else v[:, :300]) if v is not None else None for kk, v in d.items()}
    fs = _oracle(sub, steps, 0)
    assert synth.rel_frobenius(b.get(k.RAW_VEC, 0, 300), np.array([f.raw_vec() for f in fs])) <= TOL
    assert synth.rel_frobenius(b.get(k.RAW_MAT, 0, 300), np.array([f.raw_mat() for f in fs])) <= TOL
    s = ga.FilterBatch.new_ldkf(k.INFORMATION, sub["x0"], sub["P0"], sub["F"], None, sub["H"], sub["Q"], sub["R"], flags=k.FLAG_INFO_FROM_STATE | k.FLAG_STATEMENT_KERNELS)
    for t in range(steps):
        s.update(sub["y"][t])
    assert synth.rel_frobenius(b.get(k.RAW_MAT, 0, 300), s.get(k.RAW_MAT)) <= 1e-10
    assert within(synth.rel_frobenius(b.get(k.STATE, 0, 300), s.get(k.STATE)), 1e-8)   # State() = I^-1 i: a second inverse on top


@pytest.mark.parametrize("n,p,m", [(7, 2, 0), (8, 4, 1), (9, 5, 2), (10, 1, 0), (11, 7, 0), (12, 6, 0), (12, 8, 2), (12, 3, 1),
                                   (13, 2, 0), (14, 8, 1), (15, 7, 2), (16, 8, 2), (16, 6, 0), (8, 3, 2), (7, 4, 0), (8, 5, 0), (10, 4, 0), (16, 4, 1), (14, 6, 0)])
def test_infsplit_padded_family_vs_oracle(n, p, m):
    N, steps = 150, 6
    d = _model(N, n, p, m, steps, 9000 + 100 * n + 10 * p + m)
    b = ga.FilterBatch.new_ldkf(k.INFORMATION, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_INFO_FROM_STATE)
    for t in range(steps):
        b.update(d["y"][t], None if m == 0 else d["u"][t], snapshot=False)
    fs = _oracle(d, steps, m)
    assert not b.status().any()
    assert synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs])) <= TOL
    assert synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs])) <= TOL


def test_infsplit_pivoting_in_the_distributed_solve():
    """A state transition that permutes the states makes M = F^-T I F^-1 (and M + Q^-1 with a small Q^-1) far from diagonally
    dominant: the LU of the solve pivots in most columns, for some filters of a wave and not for others."""
    N, n, p, steps = 200, 12, 4, 4
    d = _model(N, n, p, 0, steps, 31337)
    rng = np.random.default_rng(5)
    for i in range(0, N, 3):   # every third filter: F = (permutation) + noise
        d["F"][i] = np.eye(n)[rng.permutation(n)] + 0.05 * rng.standard_normal((n, n))
    d["Q"] *= 1e3   # Q^-1 small against M
    b = ga.FilterBatch.new_ldkf(k.INFORMATION, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_INFO_FROM_STATE)
    for t in range(steps):
        b.update(d["y"][t], snapshot=False)
    fs = _oracle(d, steps, 0)
    assert within(synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs])), 1e-8)
    assert within(synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs])), 1e-8)


@pytest.mark.parametrize("n,p,m,awgn,from_state", [(12, 6, 0, False, True), (12, 6, 0, True, True), (9, 3, 1, True, True), (8, 4, 2, False, True),
                                                   (7, 2, 0, True, False), (16, 8, 2, False, True), (14, 5, 0, True, True), (12, 6, 0, False, False),
                                                   (11, 7, 1, False, False)])
def test_infsplit_full_estimate_vs_oracle(n, p, m, awgn, from_state):
    """KB_FLAG_FULL_ESTIMATE on the split kernels: I- and yhat = H State(prev) [+ Measurement(k)] (information.go:188-194), State(prev)
    from the distributed inverse of I (zeros while I is singular: the from_state = False cases start at i0 = 0, I0 = 0) -- the checks of
    the one-filter-per-lane kernels' test, shape for shape (yhat within the asymmetry of the oracle's own inverse, I-, I+, the
    lazily inverted PredCovariance, State())."""
    from tests.test_kinds_gpu import test_information_full_estimate_on_the_register_kernels as full_case
    full_case(n, p, m, awgn, from_state, state_rtol=1e-8)   # (State() = I^-1 i at 7..16 states: a second inverse on top of I+, as above)
