"""SRIF in fp64 on the split-lane kernel (csrc/kb_srif_split.h, round 5): one filter over 4 / 8 lanes, run-time n <= 16 and p <= 8, odd n
natively, Update (steady state and behind a Predict()), Predict(), KB_FLAG_FULL_ESTIMATE, zero-copy Phi / Htilde, a Phi that makes the
Gauss-Jordan elimination pivot, and the per-step failure path -- against the oracle (srif.go:101-160, :298-340, helper.go:142-172)."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-9
# (the kernel serves: every odd n, n < 6, 13..16 states with any p <= 8, and p = 7, 8 / 12 states with p = 5 at the even n up to 12)
SHAPES = [(1, 1), (2, 2), (3, 2), (4, 3), (5, 4), (7, 3), (7, 8), (9, 5), (11, 4), (11, 7), (13, 2), (13, 8), (14, 4), (15, 3), (15, 6), (16, 6), (16, 8), (16, 1),
          (6, 7), (8, 8), (10, 7), (12, 8), (12, 5)]


def _inputs(rng, N, n, p, steps, mix=1e-2):
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    Phi = np.eye(n) + mix * rng.standard_normal((steps, N, n, n))
    Ht = rng.standard_normal((steps, N, p, n))
    real = rng.standard_normal((steps, N, p))
    comp = real + 1e-2 * rng.standard_normal((steps, N, p))
    return x0, P0, R, Phi, Ht, real, comp


def _batch(N, n, p, x0, P0, R, flags=0):
    b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=k.F64, flags=flags)
    b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
    return b


@pytest.mark.parametrize("n,p", SHAPES)
def test_srif_split_sequences_with_full_estimates_vs_oracle(n, p):
    """Update, Update, Predict, Update (dense R), Update, Predict, Predict, Update: every mode of the kernel and every Estimate member
    (srif.go:136, :153: {Phi, b, real observation, whitened y, R, RBar}; Innovation() is the post-fit residual slot e_k here)."""
    rng = np.random.default_rng(100 * n + p)
    N, plan = 150, "uupuuppu"
    x0, P0, R, Phi, Ht, real, comp = _inputs(rng, N, n, p, len(plan))
    b = _batch(N, n, p, x0, P0, R, flags=k.FLAG_FULL_ESTIMATE)
    fs = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    for t, what in enumerate(plan):
        b.prepare(Phi[t], Ht[t])
        est = b.predict_nl() if what == "p" else b.update_nl(real[t], comp[t])
        for i, f in enumerate(fs):
            f.prepare(Phi[t, i], Ht[t, i])
            assert (f.predict_nl() if what == "p" else f.update_nl(real[t, i], comp[t, i])) == orc.OK
        eR = synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs]))
        eb = synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs]))
        eRb = synth.rel_frobenius(est.pred_covariance() if False else b.get(k.RAW_PRED_MAT), np.array([f.raw_pred_mat() for f in fs]))
        assert max(eR, eb, eRb) <= TOL, (t, what, eR, eb, eRb)
        ym = np.array([f.measurement() for f in fs])
        assert np.max(np.abs(est.measurement() - ym)) <= 1e-12 * max(1.0, np.max(np.abs(ym))), (t, what)
        assert not b.status().any()
    assert b.step() == len(plan)
    # State() / Covariance() are materialised from (b, R) by the getters' kernel: the same on both sides
    assert synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs])) <= 1e-8
    assert synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs])) <= 1e-8


@pytest.mark.parametrize("n,p", [(7, 3), (13, 5), (16, 6), (9, 8)])
def test_srif_split_pivoting_phi(n, p):
    """Phi = (a row permutation) x (1 + 5 % noise), with rows scaled so that the largest entry of a column is NOT on the diagonal for
    most columns: dgetf2's row exchanges, tracked as positions, and the rows put back in order through LDS."""
    rng = np.random.default_rng(7000 + 10 * n + p)
    N, steps = 130, 4
    x0, P0, R, Phi, Ht, real, comp = _inputs(rng, N, n, p, steps, mix=5e-2)
    for t in range(steps):
        for i in range(N):
            perm = rng.permutation(n) if i % 3 else np.roll(np.arange(n), 1 + (i % (n - 1)) if n > 1 else 0)
            Phi[t, i] = (np.diag(rng.uniform(0.5, 2.0, size=n)) @ Phi[t, i])[perm]
    b = _batch(N, n, p, x0, P0, R)
    fs = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    for t in range(steps):
        b.prepare(Phi[t], Ht[t])
        b.predict_nl() if t == 2 else b.update_nl(real[t], comp[t])
        for i, f in enumerate(fs):
            f.prepare(Phi[t, i], Ht[t, i])
            assert (f.predict_nl() if t == 2 else f.update_nl(real[t, i], comp[t, i])) == orc.OK
        eR = synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs]))
        eb = synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs]))
        assert max(eR, eb) <= TOL, (t, eR, eb)
    assert not b.status().any()


@pytest.mark.parametrize("n,p,N", [(15, 6, 1), (13, 3, 9), (16, 8, 71), (7, 2, 130), (11, 5, 257)])
def test_srif_split_zero_copy_partial_parts(n, p, N):
    """kb_prepare_dev / kb_update_nl_dev: Phi, Htilde and the observations stay in the caller's planar device arrays (leading dimension
    beyond N); batches that end inside a part (8 / 16 filters) and inside a tile."""
    import torch
    rng = np.random.default_rng(31 * n + p + N)
    steps, ld = 4, N + 5
    x0, P0, R, Phi, Ht, real, comp = _inputs(rng, N, n, p, steps)
    b = _batch(N, n, p, x0, P0, R)
    fs = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]

    def planar(M):   # [N, ...] -> [elements, ld]
        out = torch.full((int(np.prod(M.shape[1:])), ld), float("nan"), dtype=torch.float64)
        out[:, :N] = torch.from_numpy(np.ascontiguousarray(M.reshape(N, -1).T))
        return out.cuda()

    for t in range(steps):
        dPhi, dH, dr, dc = planar(Phi[t]), planar(Ht[t]), planar(real[t]), planar(comp[t])
        torch.cuda.synchronize()
        k.check(k.lib().kb_prepare_dev(b._h, dPhi.data_ptr(), dH.data_ptr(), ld))
        if t == 1:
            b.predict_nl()
        else:
            k.check(k.lib().kb_update_nl_dev(b._h, dr.data_ptr(), dc.data_ptr(), ld))
        b.synchronize()
        for i, f in enumerate(fs):
            f.prepare(Phi[t, i], Ht[t, i])
            assert (f.predict_nl() if t == 1 else f.update_nl(real[t, i], comp[t, i])) == orc.OK
    assert synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs])) <= TOL
    assert synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs])) <= TOL
    assert not b.status().any() and b.step() == steps


@pytest.mark.parametrize("n,p", [(13, 4), (16, 6), (7, 3), (11, 8)])
def test_srif_split_failures_keep_the_estimate_and_dense_leftovers_update_once(n, p):
    """srif.go:111-114 returns before anything is assigned: a filter whose Phi is singular at step k keeps (b, R) and kf.step for that
    step only.  Step 1 is a Predict(); filter 7 fails the Update behind it and keeps its DENSE R (its part stays on the dense path, one
    bit per part in the half-tile words), fails again, then succeeds; other filters fail in the steady state meanwhile."""
    rng = np.random.default_rng(4242 + n)
    N, steps = 200, 7
    x0, P0, R, Phi, Ht, real, comp = _inputs(rng, N, n, p, steps)
    fails = {2: [7], 3: [7, 40, 130], 5: [150]}
    for t, lst in fails.items():
        for i in lst:
            Phi[t, i, min(1, n - 1), :] = 0.0
    b = _batch(N, n, p, x0, P0, R)
    fs = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    nfail = np.zeros(N, dtype=np.int64)
    for t in range(steps):
        b.prepare(Phi[t], Ht[t])
        predict = t == 1
        b.predict_nl() if predict else b.update_nl(real[t], comp[t])
        for i, f in enumerate(fs):
            f.prepare(Phi[t, i], Ht[t, i])
            rc = f.predict_nl() if predict else f.update_nl(real[t, i], comp[t, i])
            bad = i in fails.get(t, [])
            assert rc == (orc.ERR_SINGULAR if bad else orc.OK)
            nfail[i] += bad
        assert synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs])) <= TOL, t
        assert synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs])) <= TOL, t
        for i in (6, 7, 8, 39, 40, 41, 130, 131, 150, 199):
            assert b.filter_step(i) == t + 1 - nfail[i], (t, i)
    assert sorted(np.nonzero(b.status())[0].tolist()) == [7, 40, 130, 150]


def test_srif_split_batch_equals_its_chunks_bit_for_bit():
    """The result of a filter may not depend on where in the grid it was computed (which part, which XCD-aware slot of the eight-lane
    mapping): 16/6 on 70 001 filters == the same data in chunks of 4 099."""
    rng = np.random.default_rng(5)
    n, p, N, C, steps = 16, 6, 70001, 4099, 2
    x0, P0, R, Phi, Ht, real, comp = _inputs(rng, N, n, p, steps)
    b = _batch(N, n, p, x0, P0, R)
    for t in range(steps):
        b.prepare(Phi[t], Ht[t]); b.update_nl(real[t], comp[t], snapshot=False)
    Rm, bv = b.get(k.RAW_MAT), b.get(k.RAW_VEC)
    for lo in range(0, N, C):
        hi = min(N, lo + C)
        c = _batch(hi - lo, n, p, x0[lo:hi], P0[lo:hi], R[lo:hi])
        for t in range(steps):
            c.prepare(Phi[t, lo:hi], Ht[t, lo:hi]); c.update_nl(real[t, lo:hi], comp[t, lo:hi], snapshot=False)
        assert np.array_equal(c.get(k.RAW_MAT), Rm[lo:hi]) and np.array_equal(c.get(k.RAW_VEC), bv[lo:hi]), lo
    f = orc.Filter.srif(x0[N - 1], P0[N - 1], R[N - 1], p)
    for t in range(steps):
        f.prepare(Phi[t, N - 1], Ht[t, N - 1]); assert f.update_nl(real[t, N - 1], comp[t, N - 1]) == orc.OK
    assert synth.rel_frobenius(Rm[N - 1:], f.raw_mat()[None]) <= TOL


@pytest.mark.parametrize("n,p,bad_filter", [(14, 4, 20), (16, 6, 52), (14, 4, 3)])
def test_srif_f32_predict_behind_a_failed_update_on_the_two_lane_kernel(n, p, bad_filter):
    """fp32 at 14 / 16 states mixes two kernels on one batch -- Predict() runs on kb_srif_split.h, Update (p <= 6) on the
    two-lane kernels -- and both keep their dense-R marks in Batch::d_srif_dense.  Predict, an Update in which a filter of slots
    16..31 / 48..63 fails (it keeps its dense RBar), Predict again (must read that filter's FULL R), Update: against the oracle."""
    rng = np.random.default_rng(777 + n + bad_filter)
    N, steps = 200, 6
    x0, P0, R, Phi, Ht, real, comp = _inputs(rng, N, n, p, steps)
    fails = {1: [bad_filter], 4: [bad_filter + 64]}
    for t, lst in fails.items():
        for i in lst:
            Phi[t, i, 1, :] = 0.0
    b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=k.F32)
    b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
    fs = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    Phi32, Ht32 = Phi.astype(np.float32).astype(np.float64), Ht.astype(np.float32).astype(np.float64)
    real32, comp32 = real.astype(np.float32).astype(np.float64), comp.astype(np.float32).astype(np.float64)
    predicts = (0, 2, 3)   # Predict, failed Update, Predict, Predict, Update (another failure, dense), Update
    for t in range(steps):
        b.prepare(Phi[t], Ht[t])
        b.predict_nl() if t in predicts else b.update_nl(real[t], comp[t])
        for i, f in enumerate(fs):
            f.prepare(Phi32[t, i], Ht32[t, i])
            rc = f.predict_nl() if t in predicts else f.update_nl(real32[t, i], comp32[t, i])
            assert rc == (orc.ERR_SINGULAR if i in fails.get(t, []) else orc.OK)
        eR = synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs]))
        eb = synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs]))
        assert eR <= 2e-5 and eb <= 2e-5, (t, eR, eb)
    assert sorted(np.nonzero(b.status())[0].tolist()) == sorted([bad_filter, bad_filter + 64])
