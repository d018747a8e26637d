"""Vanilla.Update (vanilla.go:128-220) beyond 8 states: the kernels that split ONE filter over several lanes
(gokalman_amd/csrc/kb_vanilla_split.h) against the CPU oracle, through the C ABI.

* 12 states / 6 measurements (the orbit-determination size of SURVEY 8d's config E) at 4096 filters x 20 steps, per-filter models,
  device-resident measurements: <= 1e-9 relative Frobenius on state and covariance (BASELINE.json north_star);
* the padded family: every n in 9..16, p <= 8, m <= 2, with and without KB_FLAG_FULL_ESTIMATE (all Estimate members), random dense
  H (so that the p x p factorisation pivots), control input, pure predictor, AWGN / BatchNoise replayed through the oracle;
* failure semantics: a singular innovation covariance skips that filter's step only (vanilla.go:164-167), kf.step stands still."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth
from oracle import oracle as orc
from tests.achieved import within

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _model(N, n, p, m, steps, seed):
    rng = np.random.default_rng(seed)
    F = np.eye(n) + 0.05 * rng.standard_normal((N, n, n))
    H = rng.standard_normal((N, p, n))
    A = rng.standard_normal((N, n, n))
    Q = 1e-3 * np.einsum("nij,nkj->nik", A, A) + 1e-4 * np.eye(n)
    B = rng.standard_normal((N, p, p))
    R = 1e-2 * np.einsum("nij,nkj->nik", B, B) + np.exp(rng.uniform(np.log(1e-3), np.log(1e-1), size=(N, p)))[:, :, None] * np.eye(p)
    G = rng.standard_normal((N, n, m)) if m else None
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = rng.uniform(1.0, 10.0, size=(N, n))
    y = rng.standard_normal((steps, N, p))
    u = rng.standard_normal((steps, N, m)) if m else None
    return dict(x0=x0, P0=P0, F=F, H=H, Q=Q, R=R, G=G, y=y, u=u)


def _oracle(d, kind, steps, filters=None):
    N = d["x0"].shape[0]
    out = []
    for i in (range(N) if filters is None else filters):
        f = orc.Filter.ldkf(kind, d["x0"][i], d["P0"][i], d["F"][i], None if d["G"] is None else d["G"][i], d["H"][i], d["Q"][i], d["R"][i])
        for t in range(steps):
            rc = f.update(d["y"][t, i], None if d["u"] is None else d["u"][t, i])
            assert rc == orc.OK
        out.append(f)
    return out


def test_split_12x6_at_4096_filters_20_steps_device_path_vs_oracle():
    import torch
    N, steps = 4096 + 21, 20   # (a batch that ends inside a quarter-tile)
    d = synth.linear_batch(N, 12, 6, steps)
    d["G"] = d["u"] = None
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    y_planar = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()  # [T][p][N]
    for t in range(steps):
        b.update_dev(y_planar[t].data_ptr(), N)
    b.synchronize()
    assert b.step() == steps and not b.status().any()
    xo, Po, nerr = orc.ldkf_batch(orc.VANILLA, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
    assert nerr == 0
    assert synth.rel_frobenius(b.get(k.STATE), xo) <= TOL
    assert synth.rel_frobenius(b.get(k.COVAR), Po) <= TOL
    # the same steps through kb_update_steps_dev (one register launch per step) are the same bits
    b2 = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    b2.update_steps_dev(y_planar.data_ptr(), N, steps)
    b2.synchronize()
    assert np.array_equal(b2.get(k.STATE), b.get(k.STATE)) and np.array_equal(b2.get(k.COVAR), b.get(k.COVAR))


SHAPES = [(9, 1, 0), (9, 4, 1), (10, 5, 2), (11, 7, 0), (12, 6, 0), (12, 8, 2), (12, 3, 1), (8, 6, 0), (7, 5, 2),
          (13, 2, 0), (14, 8, 1), (15, 7, 2), (16, 8, 2), (16, 6, 0), (10, 4, 0), (12, 6, 1), (16, 4, 1), (14, 6, 0)]


@pytest.mark.parametrize("n,p,m", SHAPES)
@pytest.mark.parametrize("full", [False, True])
def test_split_padded_family_vs_oracle(n, p, m, full):
    N, steps = 150, 6
    d = _model(N, n, p, m, steps, 1000 * n + 10 * p + m)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE if full else 0)
    for t in range(steps):
        est = b.update(d["y"][t], None if m == 0 else d["u"][t])
    fs = _oracle(d, orc.VANILLA, steps)
    assert not b.status().any()
    assert synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs])) <= TOL
    assert synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs])) <= TOL
    if full:
        assert synth.rel_frobenius(est.pred_covariance(), np.array([f.pred_covariance() for f in fs])) <= TOL
        assert synth.rel_frobenius(est.gain(), np.array([f.gain() for f in fs])) <= TOL
        assert within(np.max(np.abs(est.innovation() - np.array([f.innovation() for f in fs]))), 1e-8)
        assert within(np.max(np.abs(est.measurement() - np.array([f.measurement() for f in fs]))), 1e-8)


@pytest.mark.parametrize("n,p,m", [(12, 6, 0), (10, 3, 1), (16, 8, 0)])
def test_split_pure_predictor_vs_oracle(n, p, m):
    N, steps = 100, 5
    d = _model(N, n, p, m, steps, 77 + n)
    b = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE)
    for t in range(steps):
        est = b.update(d["y"][t], None if m == 0 else d["u"][t])
    fs = _oracle(d, orc.VANILLA_PREDICT, steps)
    assert synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs])) <= TOL
    assert synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs])) <= TOL
    assert synth.rel_frobenius(est.gain(), np.array([f.gain() for f in fs])) <= TOL
    assert np.max(np.abs(est.innovation())) == 0.0


@pytest.mark.parametrize("n,p", [(12, 6), (11, 4), (16, 8)])
def test_split_singular_innovation_skips_that_filter_only(n, p):
    """H = 0 and R = 0 make H P- H^T + R exactly singular: (nil, err) before kf.step++ (vanilla.go:164-167, :218); the filter's
    neighbours in the wave -- and the other lanes of the filter itself -- carry on."""
    N, steps = 200, 4
    d = _model(N, n, p, 0, steps, 5 + n)
    bad = [3, 17, 64, 199]
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    fs = [orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i]) for i in range(N)]
    Hbad, Rbad = d["H"].copy(), d["R"].copy()
    Hbad[bad] = 0.0
    Rbad[bad] = 0.0
    for t in range(steps):
        if t == 1:   # SetMeasurementMatrix / SetNoise between steps (kalman.go:44-47)
            b.set_measurement_matrix(Hbad); b.set_noise(d["Q"], Rbad)
        if t == 2:
            b.set_measurement_matrix(d["H"]); b.set_noise(d["Q"], d["R"])
        b.update(d["y"][t])
        for i, f in enumerate(fs):
            if t == 1 and i in bad:
                f.set_measurement_matrix(Hbad[i]); f.set_noise(d["Q"][i], Rbad[i])
            if t == 2 and i in bad:
                f.set_measurement_matrix(d["H"][i]); f.set_noise(d["Q"][i], d["R"][i])
            rc = f.update(d["y"][t, i])
            assert rc == (orc.ERR_SINGULAR if (t == 1 and i in bad) else orc.OK)
        st = b.status()
        assert sorted(np.nonzero(st)[0].tolist()) == (bad if t >= 1 else [])
        assert synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs])) <= TOL, t
        assert synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs])) <= TOL, t
    for i in (2, 3, 4, 64, 65):
        assert b.filter_step(i) == steps - (1 if i in bad else 0)


@pytest.mark.parametrize("n,p", [(12, 8), (14, 7), (16, 8), (11, 8)])
def test_split_distributed_inverse_row_exchanges_and_a_late_zero_pivot(n, p):
    """p = 7, 8 beyond 8 states: the p x p inverse of H P- H^T + R is formed ONCE per filter by its lanes (kb_vanilla_split.h
    dist_inverse: Gauss-Jordan by columns, pivot rows chosen by the column's owner).  R = D C D with row scales over two decades in a
    different order per filter: partial pivoting exchanges rows, differently in every lane group of a wave (scipy's LU of the first 16
    filters' S is checked to pivot in at least half of them).  Then two EQUAL rows of H with R = 0 in some filters: the zero pivot appears at the last
    column, not the first -- (nil, err) for those filters only (vanilla.go:164-167), as the oracle reports."""
    import scipy.linalg as sl
    N, steps = 200, 4
    d = _model(N, n, p, 0, steps, 31 * n + p)
    rng = np.random.default_rng(n + p)
    sc = 10.0 ** (np.array([rng.permutation(p) for _ in range(N)]) / 4.0)
    C = 0.9 * np.ones((p, p)) + 0.1 * np.eye(p)
    d["R"] = 30.0 * sc[:, :, None] * C[None] * sc[:, None, :]
    exchanging = 0   # filters of the first wave whose first-step S makes LU exchange rows
    for i in range(16):
        Pm = d["F"][i] @ d["P0"][i] @ d["F"][i].T + d["Q"][i]
        perm = np.argmax(sl.lu(d["H"][i] @ Pm @ d["H"][i].T + d["R"][i])[0], axis=0)
        exchanging += int(np.count_nonzero(perm != np.arange(p)) >= 2)
    assert exchanging >= 8
    bad = [5, 16, 17, 130, 199]
    Hbad, Rbad = d["H"].copy(), d["R"].copy()
    Hbad[bad, p - 1] = Hbad[bad, 2]
    Rbad[bad] = 0.0
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE)
    fs = [orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i]) for i in range(N)]
    for t in range(steps):
        if t == 2:
            b.set_measurement_matrix(Hbad); b.set_noise(d["Q"], Rbad)
        if t == 3:
            b.set_measurement_matrix(d["H"]); b.set_noise(d["Q"], d["R"])
        est = b.update(d["y"][t])
        for i, f in enumerate(fs):
            if t == 2 and i in bad:
                f.set_measurement_matrix(Hbad[i]); f.set_noise(d["Q"][i], Rbad[i])
            if t == 3 and i in bad:
                f.set_measurement_matrix(d["H"][i]); f.set_noise(d["Q"][i], d["R"][i])
            assert f.update(d["y"][t, i]) == (orc.ERR_SINGULAR if (t == 2 and i in bad) else orc.OK)
        assert sorted(np.nonzero(b.status())[0].tolist()) == (bad if t >= 2 else [])
        assert within(synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs])), TOL), t
        assert within(synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs])), TOL), t
        good = [i for i in range(N) if not (t == 2 and i in bad)]
        assert within(synth.rel_frobenius(est.gain()[good], np.array([fs[i].gain() for i in good])), TOL), t


@pytest.mark.parametrize("case", range(12))
def test_split_p8_structured_innovation_covariances_vs_oracle(case):
    """scripts/fuzz_split_p8.py, the first twelve cases: measurement rows that are zero / duplicated / scaled over eight decades, R with equal
    diagonal entries (ties in the pivot search), over sixteen decades, dense, or absent -- every filter the oracle fails (singular or
    ill-conditioned S) must carry a status bit and no other, the rest agree with it (the bound follows the conditioning: achieved values in
    the summary)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_split_p8", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_split_p8.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    r = mod.run_case(case)
    assert r["status_mismatches"] == 0, r
    assert r["filters_compared"] > 0
    assert within(r["worst_rel_error"], 1e-7, "%s / %s" % (r["R"], r["H"])), r


def test_split_batches_keep_the_other_kernels_bits_for_small_shapes():
    """Shapes the one-filter-per-lane kernels cover stay on them: an 8 / 4 batch run with KB_FLAG_STATEMENT_KERNELS and without agree
    to rounding, and a 12 / 6 batch equals the statement kernel to 1e-12 (different summation order, same arithmetic)."""
    N, steps = 130, 5
    d = synth.linear_batch(N, 12, 6, steps)
    a = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    s = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_STATEMENT_KERNELS)
    for t in range(steps):
        a.update(d["y"][t]); s.update(d["y"][t])
    assert synth.rel_frobenius(a.get(k.STATE), s.get(k.STATE)) <= 1e-12
    assert synth.rel_frobenius(a.get(k.COVAR), s.get(k.COVAR)) <= 1e-12


@pytest.mark.parametrize("n,p,m,full,predict", [(12, 6, 0, False, False), (12, 6, 0, True, False), (9, 2, 1, True, False), (11, 8, 2, False, False),
                                                (16, 8, 2, True, False), (14, 5, 0, False, False), (13, 3, 1, True, True), (12, 4, 0, False, True),
                                                (16, 4, 1, False, False), (10, 4, 0, True, False), (15, 6, 2, True, False), (12, 6, 1, False, False)])
def test_split_awgn_replayed_through_the_oracle(n, p, m, full, predict):
    """AWGN (noise.go:109-164) on the split kernels: the device's draws (kb_noise_sample: the standard normals of (filter, epoch,
    kf.step, which)) replayed through the oracle in the reference's call order -- Process(k) into x-, Measurement(k) into yhat,
    Process(k) again into x+ (vanilla.go:146,157,195); 4096 filters x 20 steps for 12 / 6, smaller batches for the rest."""
    bench_shape = (n, p, m) == (12, 6, 0) and not full
    N, steps = (4096, 20) if bench_shape else (150, 5)
    d = _model(N, n, p, m, steps, 31 * n + p + m)
    kind, okind = (k.VANILLA_PREDICT, orc.VANILLA_PREDICT) if predict else (k.VANILLA, orc.VANILLA)
    b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE if full else 0,
                                noise=k.NOISE_AWGN, seed=1234)
    for t in range(steps):
        est = b.update(d["y"][t], d["u"][t] if m else None, snapshot=(t == steps - 1))
    check = list(range(N)) if not bench_shape else list(range(0, N, 41)) + [N - 1]
    xs, Ps, ys, inn = [], [], [], []
    for i in check:
        LQ, LR = orc.cholesky_lower(d["Q"][i])[1], orc.cholesky_lower(d["R"][i])[1]
        f = orc.Filter.ldkf(okind, d["x0"][i], d["P0"][i], d["F"][i], d["G"][i] if m else None, d["H"][i], d["Q"][i], d["R"][i])
        for t in range(steps):
            w0, v, w2 = LQ @ b.noise_sample(i, 0, t, 0, n), LR @ b.noise_sample(i, 0, t, 1, p), LQ @ b.noise_sample(i, 0, t, 2, n)
            assert f.update(d["y"][t, i], d["u"][t, i] if m else None, w_pred=w0, v_meas=v, w_post=w2) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance()); ys.append(f.measurement()); inn.append(f.innovation())
    idx = np.array(check)
    assert synth.rel_frobenius(est.state()[idx], np.array(xs)) <= TOL
    assert synth.rel_frobenius(est.covariance()[idx], np.array(Ps)) <= TOL
    if full:
        assert synth.rel_frobenius(est.measurement()[idx], np.array(ys)) <= TOL
        if not predict:
            assert within(synth.rel_frobenius(est.innovation()[idx], np.array(inn)), 1e-7)
    assert not b.status().any() and b.step() == steps


@pytest.mark.parametrize("n,p", [(12, 3), (15, 4), (9, 2)])
def test_split_shapes_batch_noise_vs_oracle_past_n_measurements(n, p):
    """BatchNoise (noise.go:67-106): recorded vectors, the same for every filter of a batch, indexed by kf.step.  BatchNoise reports
    ZERO noise matrices (noise.go:89-98), so once steps x p >= n measurements have pinned the state down, P and S = H P- H^T are
    rounding noise and the step amplifies it without bound: measured (scripts/diag_batchnoise.py, profiles/NOTES.md round 5) NO
    evaluation keeps the oracle's digits there -- not the statement-order kernel these batches run on (the split launchers
    hand BatchNoise to it), not the reference-order register kernels at 6/3 and 8/4 -- because the problem itself has no digits: the
    oracle run on a model whose F is moved by ONE ULP per entry drifts just as far.  So the yardstick per step is that drift:
      * while the problem is well-posed (oracle-vs-perturbed-oracle <= 1e-6 for every filter): every filter within
        max(1e-9, 64 x the largest drift), state and covariance (P relative to |P0|);
      * throughout (eight steps = 24 / 32 / 16 measurements > n): the MEDIAN error within max(1e-9, 8 x the median drift), and the same
        filters fail (singular S) on both sides."""
    N, steps = 100, 8
    d = _model(N, n, p, 0, steps, 900 + n)
    rng = np.random.default_rng(n)
    proc, meas = 1e-2 * rng.standard_normal((steps, n)), 1e-2 * rng.standard_normal((steps, p))
    ZQ, ZR = np.zeros((n, n)), np.zeros((p, p))
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], ZQ, ZR, nfilters=N, flags=k.FLAG_FULL_ESTIMATE)
    b.set_batch_noise(proc, meas)
    fs = [orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], ZQ, ZR) for i in range(N)]
    Fp = np.nextafter(d["F"], np.where(rng.random(d["F"].shape) < 0.5, -np.inf, np.inf))
    fs2 = [orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], Fp[i], None, d["H"][i], ZQ, ZR) for i in range(N)]
    p0 = np.linalg.norm(d["P0"].reshape(N, -1), axis=1).max()
    well_posed_steps = 0
    for t in range(steps):
        est = b.update(d["y"][t])
        rcs = np.array([f.update(d["y"][t, i], None, proc[t], meas[t], proc[t]) for i, f in enumerate(fs)])
        for i, f in enumerate(fs2):
            f.update(d["y"][t, i], None, proc[t], meas[t], proc[t])
        st = b.status()
        assert np.array_equal(rcs != orc.OK, st != 0), (t, rcs[rcs != orc.OK], st[st != 0])
        b.clear_status()
        xs, Ps = np.array([f.state() for f in fs]), np.array([f.covariance() for f in fs])
        xd, Pd = np.array([f.state() for f in fs2]), np.array([f.covariance() for f in fs2])
        nx = np.linalg.norm(xs, axis=1)
        ex, dx = np.linalg.norm(est.state() - xs, axis=1) / nx, np.linalg.norm(xd - xs, axis=1) / nx
        eP, dP = np.abs(est.covariance() - Ps).reshape(N, -1).max(axis=1) / p0, np.abs(Pd - Ps).reshape(N, -1).max(axis=1) / p0
        print("BatchNoise %d/%d step %d: x err max %.2e median %.2e (one-ulp drift of the oracle: max %.2e median %.2e); P err max %.2e median %.2e (drift max %.2e median %.2e)"
              % (n, p, t, ex.max(), np.median(ex), dx.max(), np.median(dx), eP.max(), np.median(eP), dP.max(), np.median(dP)))
        if dx.max() <= 1e-6 and dP.max() <= 1e-6:
            well_posed_steps += 1
            assert ex.max() <= max(TOL, 64 * dx.max()) and eP.max() <= max(TOL, 64 * dP.max()), t
        assert np.median(ex) <= max(TOL, 8 * np.median(dx)) and np.median(eP) <= max(TOL, 8 * np.median(dP)), t
    assert well_posed_steps >= 2


def test_split_shared_model_one_filter_batches_and_measurement_dimension_changes():
    """The uses around the benchmark shape: (a) ONE model for all filters (every model field uploaded with broadcast = 1: the wave reads
    tile 0's model block) gives the bits of the same batch with per-filter copies; (b) a batch of ONE 12-state filter through the
    host path with every Estimate member (the drop-in use: pinned staging, FULL); (c) SetMeasurementMatrix / SetNoise with a smaller
    measurement dimension between steps (examples/jerkcar/main.go:141-159 does that every tenth step)."""
    N, n, p, steps = 200, 12, 6, 5
    d = _model(N, n, p, 0, steps, 4711)
    shared = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"][0], None, d["H"][0], d["Q"][0], d["R"][0], nfilters=N)
    rep = lambda v: np.ascontiguousarray(np.broadcast_to(v[0], v.shape))
    perf = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], rep(d["F"]), None, rep(d["H"]), rep(d["Q"]), rep(d["R"]))
    for t in range(steps):
        shared.update(d["y"][t]); perf.update(d["y"][t])
    assert np.array_equal(shared.get(k.STATE), perf.get(k.STATE)) and np.array_equal(shared.get(k.COVAR), perf.get(k.COVAR))
    # (b) + (c): one filter, p = 6 -> 3 -> 6
    i = 17
    one = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i], flags=k.FLAG_FULL_ESTIMATE, pmax=p)
    f = orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
    H3, R3 = d["H"][i][:3], d["R"][i][:3, :3]
    for t in range(steps):
        if t == 2:
            one.set_measurement_matrix(H3); one.set_noise(d["Q"][i], R3)
            f.set_measurement_matrix(H3); f.set_noise(d["Q"][i], R3)
        if t == 4:
            one.set_measurement_matrix(d["H"][i]); one.set_noise(d["Q"][i], d["R"][i])
            f.set_measurement_matrix(d["H"][i]); f.set_noise(d["Q"][i], d["R"][i])
        y = d["y"][t, i][:3] if t in (2, 3) else d["y"][t, i]
        est = one.update(y)
        assert f.update(y) == orc.OK
        assert synth.rel_frobenius(est.state().reshape(1, -1), f.state().reshape(1, -1)) <= TOL, t
        assert synth.rel_frobenius(est.covariance().reshape(1, -1), f.covariance().reshape(1, -1)) <= TOL, t
        assert synth.rel_frobenius(est.pred_covariance().reshape(1, -1), f.pred_covariance().reshape(1, -1)) <= TOL, t
        assert synth.rel_frobenius(est.gain().reshape(1, -1), f.gain().reshape(1, -1)) <= TOL, t
        assert np.max(np.abs(est.innovation().ravel() - f.innovation())) <= 1e-9 and np.max(np.abs(est.measurement().ravel() - f.measurement())) <= 1e-9
    assert one.step() == steps


@pytest.mark.parametrize("kind", [k.VANILLA, k.SQUAREROOT, k.INFORMATION])
@pytest.mark.parametrize("n,p", [(12, 6), (16, 4), (14, 7), (10, 4)])
def test_split_sharded_batch_is_bit_equal_to_the_unsharded_batch(kind, n, p):
    """Filters share nothing (vanilla.go:216-218): the same 4099 filters as ONE batch and as three shards (kb_sharded_*: ragged
    boundaries [g N / G, (g + 1) N / G), so parts and -- with eight lanes per filter -- the XCD-aware part mapping
    (split_part_of_block) fall differently in each) are the same bits, and every filter was stepped exactly once per call."""
    N, steps, shards = 4099, 4, 3
    d = _model(N, n, p, 0, steps, 4242 + n)
    flags = k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0
    one = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=flags)
    sh = ga.ShardedBatch(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], N, devices=[0] * shards, flags=flags)
    for t in range(steps):
        one.update(d["y"][t], snapshot=False)
        sh.update(d["y"][t])
    raw_v, raw_m = (k.RAW_VEC, k.RAW_MAT) if kind == k.INFORMATION else (k.STATE, k.COVAR)
    assert np.array_equal(sh.get(raw_v, (n,)), one.get(raw_v))
    assert np.array_equal(sh.get(raw_m, (n, n)), one.get(raw_m))
    assert not sh.status().any() and not one.status().any()
    assert all(one.filter_step(i) == steps for i in (0, 63, 64, 2047, 4098))
