"""SRIF (srif.go:101-160, :298-340) at the benchmark size and through its failure path.

* config E at size: 262 144 filters x 12/6 (fp32 and fp64), inputs from a seeded device generator, handed over zero-copy
  (kb_prepare_dev / kb_update_nl_dev).  Sampled filters (first tile, last tile, random tiles) are checked against the
  oracle, and the WHOLE batch must be bit-equal to the same data run in 4096-filter chunks: the result of a filter may not
  depend on where in the grid (which workgroup, which wave, which LDS region, which round of waves) it was computed.  (The
  round-1 one-filter-per-lane kernels, still reachable with KB_SRIF_ONE_LANE=1, are a persistent grid with an LDS double
  buffer and a flag hand-over that only wraps around at this size.)
* per-step failure semantics (srif.go:111-114 returns before anything is assigned): a filter whose Phi is singular at
  step k keeps its estimate for that step only, on every SRIF kernel path."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
# fp32 SRIF against the fp64 oracle: achieved 1.3e-6 (R) / 2.7e-6 (b) over 4096 filters x 20 Updates (bench.py extra.srif_fp32.parity prints
# it on every run); the bound is a little under 10x that -- the sequences here contain Predict() steps and 2^18 filters
SRIF_F32_TOL = 2e-5


def _srif_batch(N, n, p, dtype, x0, P0, R, flags=0):
    b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=dtype, flags=flags)
    b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
    return b


@pytest.mark.parametrize("dtype,tol", [(k.F32, SRIF_F32_TOL), (k.F64, 1e-9)])
def test_srif_config_e_at_size_vs_oracle_and_chunked(dtype, tol):
    import torch
    N, n, p, CH = 1 << 18, 12, 6, 4096
    tdt = torch.float32 if dtype == k.F32 else torch.float64
    g = torch.Generator(device="cuda"); g.manual_seed(20260)
    rng = np.random.default_rng(17)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10.0] * 6 + [1.0] * 6
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    # sequence: Update (two-lane kernel, triangular R), Predict (time kernel: R becomes the dense RBar), Update (the DENSE
    # variant of the two-lane kernel), Update (steady state again)
    SEQ = ["update", "predict", "update", "update"]
    eye = torch.eye(n, dtype=tdt, device="cuda").reshape(n * n, 1)
    Phi = [(eye + 1e-2 * torch.randn(n * n, N, dtype=tdt, device="cuda", generator=g)).contiguous() for _ in SEQ]
    Ht = [torch.randn(p * n, N, dtype=tdt, device="cuda", generator=g) for _ in SEQ]
    real = [torch.randn(p, N, dtype=tdt, device="cuda", generator=g) for _ in SEQ]
    comp = [(real[t] + 1e-2 * torch.randn(p, N, dtype=tdt, device="cuda", generator=g)).contiguous() for t in range(len(SEQ))]
    torch.cuda.synchronize()   # the handles' streams do not wait for torch's
    es = 4 if dtype == k.F32 else 8

    def drive(b, first, count):
        for t, what in enumerate(SEQ):
            k.check(k.lib().kb_prepare_dev(b._h, Phi[t].data_ptr() + first * es, Ht[t].data_ptr() + first * es, N))
            if what == "predict":
                b.predict_nl()
            else:
                k.check(k.lib().kb_update_nl_dev(b._h, real[t].data_ptr() + first * es, comp[t].data_ptr() + first * es, N))
        b.synchronize()

    big = _srif_batch(N, n, p, dtype, x0, P0, R)
    drive(big, 0, N)
    assert not big.status().any()
    bvec, bmat = big.get(k.RAW_VEC), big.get(k.RAW_MAT)
    assert np.isfinite(bvec).all() and np.isfinite(bmat).all()

    # the whole batch against 4096-filter chunks: bit for bit
    for c0 in range(0, N, CH):
        small = _srif_batch(CH, n, p, dtype, x0[c0:c0 + CH], P0[c0:c0 + CH], R[c0:c0 + CH])
        drive(small, c0, CH)
        assert np.array_equal(small.get(k.RAW_VEC), bvec[c0:c0 + CH]), "b differs in chunk at %d" % c0
        assert np.array_equal(small.get(k.RAW_MAT), bmat[c0:c0 + CH]), "R differs in chunk at %d" % c0
        small.close()

    # sampled filters against the oracle (fp64, reference order)
    tiles = np.concatenate([[0, N // 64 - 1], rng.choice(N // 64, size=6, replace=False)])
    idx = np.unique(np.concatenate([np.arange(t * 64, t * 64 + 64) for t in tiles]))
    assert idx.size == 512
    ti = torch.from_numpy(idx).cuda()
    hPhi = [P_[:, ti].T.double().cpu().numpy().reshape(-1, n, n) for P_ in Phi]
    hHt = [H_[:, ti].T.double().cpu().numpy().reshape(-1, p, n) for H_ in Ht]
    hre = [r_[:, ti].T.double().cpu().numpy() for r_ in real]
    hco = [c_[:, ti].T.double().cpu().numpy() for c_ in comp]
    bs, Rs = [], []
    for j, i in enumerate(idx):
        f = orc.Filter.srif(x0[i], P0[i], R[i], p)
        for t, what in enumerate(SEQ):
            f.prepare(hPhi[t][j], hHt[t][j])
            assert (f.predict_nl() if what == "predict" else f.update_nl(hre[t][j], hco[t][j])) == orc.OK
        bs.append(f.raw_vec()); Rs.append(f.raw_mat())
    assert synth.rel_frobenius(bmat[idx], np.array(Rs)) <= tol
    assert synth.rel_frobenius(bvec[idx], np.array(bs)) <= tol


@pytest.mark.parametrize("fail_step", [1, 3])   # 1: steady state (triangular R); 3: right after a Predict() (dense R: the skipped filters keep it)
@pytest.mark.parametrize("n,p,dtype,tol", [(12, 6, k.F32, SRIF_F32_TOL), (12, 6, k.F64, 1e-9), (6, 2, k.F64, 1e-9), (6, 2, k.F32, SRIF_F32_TOL), (5, 2, k.F64, 1e-9),
                                           (8, 2, k.F64, 1e-9), (8, 4, k.F32, SRIF_F32_TOL), (10, 2, k.F64, 1e-9), (10, 4, k.F64, 1e-9), (10, 2, k.F32, SRIF_F32_TOL),
                                           (12, 2, k.F64, 1e-9), (12, 4, k.F32, SRIF_F32_TOL), (12, 4, k.F64, 1e-9),
                                           (6, 1, k.F64, 1e-9), (6, 3, k.F64, 1e-9), (6, 4, k.F32, SRIF_F32_TOL), (8, 1, k.F32, SRIF_F32_TOL), (8, 3, k.F64, 1e-9), (10, 3, k.F64, 1e-9),
                                           (10, 1, k.F64, 1e-9), (12, 1, k.F64, 1e-9), (12, 3, k.F32, SRIF_F32_TOL), (12, 3, k.F64, 1e-9),
                                           (12, 5, k.F64, 1e-9), (12, 5, k.F32, SRIF_F32_TOL),
                                           (7, 3, k.F64, 1e-9), (9, 2, k.F64, 1e-9), (11, 4, k.F32, SRIF_F32_TOL), (11, 6, k.F64, 1e-9), (5, 1, k.F64, 1e-9), (11, 5, k.F64, 1e-9), (13, 3, k.F64, 1e-9),
                                           (8, 6, k.F64, 1e-9), (10, 5, k.F32, SRIF_F32_TOL), (6, 6, k.F64, 1e-9), (9, 5, k.F64, 1e-9), (3, 2, k.F64, 1e-9), (4, 4, k.F32, SRIF_F32_TOL),
                                           (8, 8, k.F64, 1e-9), (10, 7, k.F64, 1e-9), (6, 8, k.F32, SRIF_F32_TOL), (12, 8, k.F32, SRIF_F32_TOL), (7, 7, k.F64, 1e-9), (12, 7, k.F64, 1e-9), (11, 8, k.F64, 1e-9),
                                           (14, 4, k.F64, 1e-9), (16, 6, k.F64, 1e-9), (16, 3, k.F32, SRIF_F32_TOL), (15, 2, k.F64, 1e-9), (14, 7, k.F64, 1e-9)])
def test_srif_singular_phi_skips_only_that_step(n, p, dtype, tol, fail_step):
    """(14, 7) has no register kernel (more than six measurements beyond 12 states): the generic one must behave the same; 13 .. 16 states
    Predict() on the generic kernel between Updates on the two-lane one; 8 / 10 / 12 states with 2 / 4 measurements run further
    instantiations of the two-lanes-per-filter kernel (kb_srif_pair*b.hip, *c.hip) and of the Predict() kernel; an odd number of measurements runs
    the next even instantiation with a padded row (kb_srif_pair.h PADM); an odd number of states (5 with p != 2, 7, 9, 11) the next even
    one as diag(filter, one uncoupled state) (round 4: kb_srif_odd.hip; round 5: natively on kb_srif_split.h)."""
    rng = np.random.default_rng(100 * n + fail_step)
    N, steps = 130, 6
    bad = [7, 70, 129]
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    Phi = np.eye(n) + 1e-2 * rng.standard_normal((steps, N, n, n))
    Ht = rng.standard_normal((steps, N, p, n))
    real = rng.standard_normal((steps, N, p))
    comp = real + 1e-2 * rng.standard_normal((steps, N, p))
    for i in bad:
        Phi[fail_step, i, 2, :] = 0.0    # exactly singular: gonum's Inverse returns an error (srif.go:112-114)
    b = _srif_batch(N, n, p, dtype, x0, P0, R)
    filters = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    for t in range(steps):
        b.prepare(Phi[t], Ht[t])
        predict = t == 2
        if predict:
            b.predict_nl()
        else:
            b.update_nl(real[t], comp[t])
        for i, f in enumerate(filters):
            f.prepare(Phi[t, i], Ht[t, i])
            rc = f.predict_nl() if predict else f.update_nl(real[t, i], comp[t, i])
            assert rc == (orc.ERR_SINGULAR if (t == fail_step and i in bad) else orc.OK)
        st = b.status()
        if t < fail_step:
            assert not st.any()
        else:   # a sticky report, not a gate
            assert sorted(np.nonzero(st)[0].tolist()) == bad and set(st[bad].tolist()) == {k.ST_SINGULAR}
        if t >= fail_step:   # the failing step itself (estimate untouched), then k+1, k+2 ... run normally
            assert synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in filters])) <= tol, t
            assert synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in filters])) <= tol, t


@pytest.mark.parametrize("N", [1, 31, 33, 65])
@pytest.mark.parametrize("dtype,tol", [(k.F32, SRIF_F32_TOL), (k.F64, 1e-9)])
def test_srif_partial_half_tiles_zero_copy_and_full_estimate(N, dtype, tol):
    """The two-lanes-per-filter kernel owns 32 filters per wave: batches that end inside a half-tile, models read in place from
    planar arrays whose leading dimension exceeds N (kb_prepare_dev), every Estimate member written (FULL_ESTIMATE)."""
    import torch
    n, p, ld, steps = 12, 6, 200, 3
    rng = np.random.default_rng(1000 + N)
    tdt = torch.float32 if dtype == k.F32 else torch.float64
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10.0] * 6 + [1.0] * 6
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    b = _srif_batch(N, n, p, dtype, x0, P0, R, flags=k.FLAG_FULL_ESTIMATE)
    filters = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    for t in range(steps):
        Phi = np.eye(n) + 1e-2 * rng.standard_normal((N, n, n))
        Ht = rng.standard_normal((N, p, n))
        real = rng.standard_normal((N, p)); comp = real + 1e-2 * rng.standard_normal((N, p))
        dPhi = torch.full((n * n, ld), float("nan"), dtype=tdt, device="cuda"); dPhi[:, :N] = torch.from_numpy(Phi.reshape(N, -1).T.copy()).to(tdt)
        dH = torch.full((p * n, ld), float("nan"), dtype=tdt, device="cuda"); dH[:, :N] = torch.from_numpy(Ht.reshape(N, -1).T.copy()).to(tdt)
        dre = torch.full((p, ld), float("nan"), dtype=tdt, device="cuda"); dre[:, :N] = torch.from_numpy(real.T.copy()).to(tdt)
        dco = torch.full((p, ld), float("nan"), dtype=tdt, device="cuda"); dco[:, :N] = torch.from_numpy(comp.T.copy()).to(tdt)
        torch.cuda.synchronize()   # (fills and copies run on torch's stream; the handle's does not wait for it)
        k.check(k.lib().kb_prepare_dev(b._h, dPhi.data_ptr(), dH.data_ptr(), ld))
        k.check(k.lib().kb_update_nl_dev(b._h, dre.data_ptr(), dco.data_ptr(), ld))
        b.synchronize()
        # the oracle sees what the device saw (fp32 inputs are rounded on the way in)
        Phi_d = dPhi[:, :N].T.double().cpu().numpy().reshape(N, n, n); Ht_d = dH[:, :N].T.double().cpu().numpy().reshape(N, p, n)
        re_d = dre[:, :N].T.double().cpu().numpy(); co_d = dco[:, :N].T.double().cpu().numpy()
        for i, f in enumerate(filters):
            f.prepare(Phi_d[i], Ht_d[i])
            assert f.update_nl(re_d[i], co_d[i]) == orc.OK
    est = b.estimate(snapshot=True)
    assert not est.status().any()
    assert synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in filters])) <= tol
    assert synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in filters])) <= tol
    assert synth.rel_frobenius(est.state(), np.array([f.state() for f in filters])) <= tol * 10
    assert synth.rel_frobenius(est.pred_covariance(), np.array([f.pred_covariance() for f in filters])) <= tol * 10
    assert synth.rel_frobenius(est.measurement(), np.array([f.measurement() for f in filters])) <= tol
    assert np.isfinite(est.covariance()).all()


@pytest.mark.parametrize("n,p,dtype,tol", [(12, 6, k.F64, 1e-9), (12, 6, k.F32, SRIF_F32_TOL), (6, 2, k.F64, 1e-9)])
def test_srif_leftover_dense_tiles_and_a_new_failure_update_every_filter_once(n, p, dtype, tol):
    """while some filter may still hold a dense R (it failed the Update that followed a Predict()), every Update
    launches the steady-state kernel AND the dense kernel.  Each half-tile must be taken by exactly one of them, also when a filter
    of an untouched half-tile fails in the steady-state kernel of that very step (its 31 neighbours must not get the measurement
    twice, its kf.step must fall behind by one, not two)."""
    rng = np.random.default_rng(4242 + n)
    N, steps = 200, 7
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    Phi = np.eye(n) + 1e-2 * rng.standard_normal((steps, N, n, n))
    Ht = rng.standard_normal((steps, N, p, n))
    real = rng.standard_normal((steps, N, p))
    comp = real + 1e-2 * rng.standard_normal((steps, N, p))
    # step 1: Predict().  step 2: filter 7 fails in the whole-batch dense Update (keeps its dense R: half-tile 0 stays with the
    # dense kernel).  step 3: filters 40 (half-tile 1) and 130 (tile 2) fail in the steady-state kernel, filter 7 fails AGAIN in the
    # dense one.  step 4: everybody succeeds (filter 7 in the dense kernel).  steps 5, 6: steady state; 150 fails at 5.
    fails = {2: [7], 3: [7, 40, 130], 5: [150]}
    for t, lst in fails.items():
        for i in lst:
            Phi[t, i, 1, :] = 0.0
    b = _srif_batch(N, n, p, dtype, x0, P0, R)
    filters = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    nfail = np.zeros(N, dtype=np.int64)
    for t in range(steps):
        b.prepare(Phi[t], Ht[t])
        predict = t == 1
        if predict:
            b.predict_nl()
        else:
            b.update_nl(real[t], comp[t])
        for i, f in enumerate(filters):
            f.prepare(Phi[t, i], Ht[t, i])
            rc = f.predict_nl() if predict else f.update_nl(real[t, i], comp[t, i])
            bad = i in fails.get(t, [])
            assert rc == (orc.ERR_SINGULAR if bad else orc.OK)
            nfail[i] += bad
        assert synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in filters])) <= tol, t
        assert synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in filters])) <= tol, t
        for i in (6, 7, 8, 39, 40, 41, 130, 131, 150, 199):   # kf.step = calls - failed calls, per filter
            assert b.filter_step(i) == t + 1 - nfail[i], (t, i)
    st = b.status()
    assert sorted(np.nonzero(st)[0].tolist()) == [7, 40, 130, 150]


@pytest.mark.parametrize("n,p,dtype,tol", [(7, 3, k.F64, 1e-9), (11, 4, k.F64, 1e-9), (9, 2, k.F32, SRIF_F32_TOL)])
def test_srif_odd_states_shadow_follows_every_other_writer_of_the_state(n, p, dtype, tol):
    """(Written for round 4's kb_srif_odd.hip, which kept a widened copy of the state between consecutive steps; the odd shapes now run
    natively on kb_srif_split.h and the sequence stays as a regression test.)  kb_reset and a step that fails for some filters
    have to show in the step after them, and a second handle has its own copy."""
    rng = np.random.default_rng(4242 + n)
    N, steps = 100, 3
    x0 = rng.standard_normal((N, n)); x1 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = rng.uniform(1.0, 10.0, size=(N, n))
    P1 = np.zeros((N, n, n)); P1[:, np.arange(n), np.arange(n)] = rng.uniform(0.5, 2.0, size=(N, n))
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    Phi = np.eye(n) + 1e-2 * rng.standard_normal((3 * steps, N, n, n))
    Ht = rng.standard_normal((3 * steps, N, p, n))
    real = rng.standard_normal((3 * steps, N, p)); comp = real + 1e-2 * rng.standard_normal((3 * steps, N, p))
    Phi[1, 5, 2, :] = 0.0   # one filter skips step 1 (singular Phi): its state stays, in the shadow as well
    b = _srif_batch(N, n, p, dtype, x0, P0, R)

    def run(t0, xs, Ps):
        fs = [orc.Filter.srif(xs[i], Ps[i], R[i], p) for i in range(N)]
        for t in range(t0, t0 + steps):
            b.prepare(Phi[t], Ht[t])
            b.predict_nl() if t % steps == 2 else b.update_nl(real[t], comp[t])
            for i, f in enumerate(fs):
                f.prepare(Phi[t, i], Ht[t, i])
                f.predict_nl() if t % steps == 2 else f.update_nl(real[t, i], comp[t, i])
        assert synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs])) <= tol
        assert synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs])) <= tol

    run(0, x0, P0)
    assert b.status()[5] == k.ST_SINGULAR
    b.reset()
    run(steps, x0, P0)                 # Reset(): back to the constructor's estimate
    with pytest.raises(ga.KalmanError, match="constructor arguments"):   # (x0 / P0 cannot change after kb_init: no third writer)
        b.set(k.X, x1, 1)
    b2 = _srif_batch(N, n, p, dtype, x1, P1, R)   # a second handle has its own shadow
    b, keep = b2, b
    run(2 * steps, x1, P1)


@pytest.mark.parametrize("N,fail", [(4096, False), (1000, True), (70, True)])
def test_srif_time_fused_steps_equal_single_steps_bit_for_bit(N, fail):
    _fused_against_single_steps(N, fail, 12, 6)


@pytest.mark.parametrize("N,fail", [(1000, True), (130, False)])
def test_srif_time_fused_6x2_the_references_own_shape(N, fail):
    _fused_against_single_steps(N, fail, 6, 2)


def _fused_against_single_steps(N, fail, n, p):
    """kb_update_nl_steps_dev (round 6): T Prepare + Update pairs of config E's shape (12 / 6, fp32, zero-copy operands, steady state) in ONE
    launch, the rows of (b, R) resident in registers between the steps -- the same operations in the same order as T single calls: the same
    bits, kf.step and the per-step failure semantics included (a filter whose Phi is singular at step k skips that step only; its wave
    reloads from memory at the next).  Also from a batch that is NOT in the steady state (behind a Predict(): the call falls back to T
    launches), and for a shape without a fused kernel."""
    import torch
    T = 7
    rng = np.random.default_rng(31 + N)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10.0] * (n // 2) + [1.0] * (n // 2)
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
    ld = N + 13                                   # (a leading dimension beyond N: the planar arrays' own)
    Phi = np.eye(n) + 1e-2 * rng.standard_normal((T, N, n, n))
    if fail:
        Phi[2, 5, 3, :] = 0.0; Phi[2, N - 1, 0, :] = 0.0; Phi[5, 5, 1, :] = 0.0      # filter 5 fails steps 2 and 5, the last filter step 2
    Ht = rng.standard_normal((T, N, p, n)); real = rng.standard_normal((T, N, p)); comp = real + 1e-2 * rng.standard_normal((T, N, p))

    def planar(a):                                # [T][N][...] -> device [T][elems][ld] fp32, NaN behind N
        out = torch.full((T, int(np.prod(a.shape[2:])), ld), float("nan"), dtype=torch.float32)
        out[:, :, :N] = torch.from_numpy(a.reshape(T, N, -1).transpose(0, 2, 1).astype(np.float32))
        return out.cuda()
    dPhi, dH, dre, dco = planar(Phi), planar(Ht), planar(real), planar(comp)
    torch.cuda.synchronize()
    res = []
    for fused in (True, False):
        b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=k.F32)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
        if fused:
            b.update_nl_steps_dev(dPhi.data_ptr(), dH.data_ptr(), ld, n * n * ld, p * n * ld, dre.data_ptr(), dco.data_ptr(), ld, p * ld, T)
            assert "srif_pair_fused_kernel<float, %d, %d>" % (n, p) in b.last_kernel()
        else:
            for t in range(T):
                k.check(k.lib().kb_prepare_dev(b._h, dPhi[t].data_ptr(), dH[t].data_ptr(), ld))
                k.check(k.lib().kb_update_nl_dev(b._h, dre[t].data_ptr(), dco[t].data_ptr(), ld))
        b.synchronize()
        res.append((b.get(k.RAW_MAT), b.get(k.RAW_VEC), b.status().copy(), [b.filter_step(i) for i in (4, 5, 6, N - 1)], b.calls()))
    assert np.array_equal(res[0][0].view(np.uint64), res[1][0].view(np.uint64)) and np.array_equal(res[0][1].view(np.uint64), res[1][1].view(np.uint64))
    assert np.array_equal(res[0][2], res[1][2]) and res[0][3] == res[1][3] == ([T, T - 2, T, T - 1] if fail else [T] * 4) and res[0][4] == res[1][4] == T
    if not fail:
        assert not res[0][2].any()
    else:
        assert sorted(np.nonzero(res[0][2])[0].tolist()) == [5, N - 1]
    # not in the steady state (a Predict() pending a dense R), and a shape without a fused kernel: T launches from the one call, same results as the loop
    if n != 12:
        return
    for shape, predict_first in (((12, 6), True), ((8, 4), False)):
        nn, pp = shape
        outs = []
        for fused in (True, False):
            b = ga.FilterBatch(k.SRIF, nn, pp, 0, N, dtype=k.F32)
            b.set(k.X, x0[:, :nn], 1); b.set(k.P, P0[:, :nn, :nn], 2); b.set(k.R, R[:, :pp, :pp], 2, p_rows=pp); b.init()
            sub = lambda a, r, c: np.ascontiguousarray(a[:, :, :r, :c]) if a.ndim == 4 else np.ascontiguousarray(a[:, :, :r])
            qPhi, qH, qre, qco = planar(sub(Phi, nn, nn)), planar(sub(Ht, pp, nn)), planar(sub(real, pp, 0)), planar(sub(comp, pp, 0))
            torch.cuda.synchronize()
            if predict_first:
                k.check(k.lib().kb_prepare_dev(b._h, qPhi[0].data_ptr(), qH[0].data_ptr(), ld)); b.predict_nl(snapshot=False)
            if fused:
                b.update_nl_steps_dev(qPhi.data_ptr(), qH.data_ptr(), ld, nn * nn * ld, pp * nn * ld, qre.data_ptr(), qco.data_ptr(), ld, pp * ld, T)
            else:
                for t in range(T):
                    k.check(k.lib().kb_prepare_dev(b._h, qPhi[t].data_ptr(), qH[t].data_ptr(), ld))
                    k.check(k.lib().kb_update_nl_dev(b._h, qre[t].data_ptr(), qco[t].data_ptr(), ld))
            b.synchronize()
            outs.append((b.get(k.RAW_MAT), b.get(k.RAW_VEC), b.status().copy()))
        assert np.array_equal(outs[0][0].view(np.uint64), outs[1][0].view(np.uint64)) and np.array_equal(outs[0][1].view(np.uint64), outs[1][1].view(np.uint64))
        assert np.array_equal(outs[0][2], outs[1][2])
