"""Every shape of the north star's envelope once: n = 1..16 states x p = 1..8 measurements (m = 0, 1, 2 controls and
KB_FLAG_FULL_ESTIMATE alternating with the shape), Vanilla / SquareRoot / Information / HybridKF / SRIF, a batch that ends inside a tile,
three steps, against the CPU oracle.  The other GPU tests pick shapes per kernel; this one is about the DISPATCH -- whichever kernel a
shape lands on (exact, padded, split-lane, statement), the result is the reference's."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth
from oracle import oracle as orc
from tests.achieved import within

pytestmark = pytest.mark.gpu
N, STEPS = 70, 3


def _model(n, p, m, seed, N=N, STEPS=STEPS):
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
    y = rng.standard_normal((STEPS, N, p))
    u = rng.standard_normal((STEPS, N, m)) if m else None
    return dict(x0=x0, P0=P0, F=F, H=H, Q=Q, R=R, G=G, y=y, u=u)


GRID = [(n, p) for n in range(1, 17) for p in range(1, 9)]


@pytest.mark.parametrize("kind,okind", [(k.VANILLA, orc.VANILLA), (k.SQUAREROOT, orc.SQUAREROOT)])
def test_every_shape_vanilla_squareroot(kind, okind):
    bad = []
    for n, p in GRID:
        m, full = (n + p) % 3, bool((n + p) % 2)
        d = _model(n, p, m, 100 * n + p)
        b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE if full else 0)
        fs = [orc.Filter.ldkf(okind, d["x0"][i], d["P0"][i], d["F"][i], None if m == 0 else d["G"][i], d["H"][i], d["Q"][i], d["R"][i]) for i in range(N)]
        for t in range(STEPS):
            est = b.update(d["y"][t], None if m == 0 else d["u"][t])
            for i, f in enumerate(fs):
                assert f.update(d["y"][t, i], None if m == 0 else d["u"][t, i]) == orc.OK
        ex = synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs]))
        eP = synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs]))
        eK = synth.rel_frobenius(est.gain(), np.array([f.gain() for f in fs])) if full else 0.0
        ePm = synth.rel_frobenius(est.pred_covariance(), np.array([f.pred_covariance() for f in fs])) if full else 0.0
        if b.status().any() or b.step() != STEPS or max(ex, eP, eK, ePm) > 1e-9:
            bad.append((n, p, m, full, ex, eP, eK, ePm))
    assert not bad, bad


def test_every_shape_information():
    bad = []
    for n, p in GRID:
        m = (n + p) % 3
        d = _model(n, p, m, 300 * n + p)
        b = ga.FilterBatch.new_ldkf(k.INFORMATION, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_INFO_FROM_STATE)
        fs = [orc.Filter.information_from_state(d["x0"][i], d["P0"][i], d["F"][i], None if m == 0 else d["G"][i], d["H"][i], d["Q"][i], d["R"][i]) for i in range(N)]
        for t in range(STEPS):
            b.update(d["y"][t], None if m == 0 else d["u"][t], snapshot=False)
            for i, f in enumerate(fs):
                assert f.update(d["y"][t, i], None if m == 0 else d["u"][t, i]) == orc.OK
        ei = synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs]))
        eI = synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs]))
        if b.status().any() or b.step() != STEPS or max(ei, eI) > 1e-9:
            bad.append((n, p, m, ei, eI))
    assert not bad, bad


def _nl(n, p, rng):
    Phi = np.eye(n) + 1e-2 * rng.standard_normal((STEPS, N, n, n))
    Ht = rng.standard_normal((STEPS, N, p, n))
    real = rng.standard_normal((STEPS, N, p))
    comp = real + 1e-2 * rng.standard_normal((STEPS, N, p))
    return Phi, Ht, real, comp


@pytest.mark.parametrize("ekf", [False, True])
def test_every_shape_hybrid(ekf):
    bad = []
    for n, p in GRID:
        rng = np.random.default_rng(500 * n + p)
        full = bool((n + p) % 2)
        x0 = rng.standard_normal((N, n))
        P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = rng.uniform(1.0, 10.0, size=(N, n))
        R = np.tile(np.diag(np.full(p, 1e-2)), (N, 1, 1))
        Phi, Ht, real, comp = _nl(n, p, rng)
        b = ga.FilterBatch(k.HYBRID, n, p, 0, N, flags=k.FLAG_FULL_ESTIMATE if full else 0)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
        if ekf:
            b.enable_ekf()
        fs = []
        for i in range(N):
            f = orc.Filter.hybrid(x0[i], P0[i], None, R[i], p)
            if ekf:
                f.enable_ekf()
            fs.append(f)
        for t in range(STEPS):
            b.prepare(Phi[t], Ht[t])
            est = b.predict_nl() if t == 1 else b.update_nl(real[t], comp[t])
            for i, f in enumerate(fs):
                f.prepare(Phi[t, i], Ht[t, i])
                assert (f.predict_nl() if t == 1 else f.update_nl(real[t, i], comp[t, i])) == orc.OK
        ex = synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs]))
        eP = synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs]))
        eK = synth.rel_frobenius(est.gain(), np.array([f.gain() for f in fs])) if full else 0.0
        # (p > n with R = 1e-2 I: the update cancels x- against K (y - H x-) to ~1e-7 of its size; the covariance and the gain stay at 1e-15)
        if b.status().any() or b.step() != STEPS or max(eP, eK) > 1e-9 or not within(ex, 1e-9 if p <= n else 1e-7, "hybrid %d/%d state" % (n, p)):
            bad.append((n, p, full, ex, eP, eK))
    assert not bad, bad


@pytest.mark.parametrize("n,p", [(12, 8), (16, 8), (14, 7), (9, 8)])
@pytest.mark.parametrize("ekf", [False, True])
def test_hybrid_p8_row_exchanges_and_singular_innovation(n, p, ekf):
    """HybridKF at 7, 8 measurements beyond 8 states (the Vanilla split kernel's HYB mode: S^-1 once per filter by its lanes).  R = D C D with row
    scales over two decades in a different order per filter (the p x p inverse exchanges rows, differently in the lane groups of a wave); a third
    of the filters have a noise-free measurement 3 whose row of Htilde is zero at the last step: H P H^T + R is exactly singular -- hybrid.go:149-152 returns the
    error before kf.step++, the others carry on."""
    rng = np.random.default_rng(40 * n + p + int(ekf))
    M = 96
    x0 = rng.standard_normal((M, n))
    P0 = np.zeros((M, n, n)); P0[:, np.arange(n), np.arange(n)] = rng.uniform(1.0, 10.0, size=(M, n))
    sc = 10.0 ** (np.array([rng.permutation(p) for _ in range(M)]) / 4.0)
    C = 0.9 * np.ones((p, p)) + 0.1 * np.eye(p)
    R = 0.3 * sc[:, :, None] * C[None] * sc[:, None, :]
    bad = sorted(rng.choice(M, size=M // 3, replace=False).tolist())
    R[bad, 3, :] = 0.0; R[bad, :, 3] = 0.0   # (HybridKF takes its noise at construction, hybrid.go:21-47: measurement 3 of these filters is noise-free)
    b = ga.FilterBatch(k.HYBRID, n, p, 0, M, flags=k.FLAG_FULL_ESTIMATE)
    b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
    fs = [orc.Filter.hybrid(x0[i], P0[i], None, R[i], p) for i in range(M)]
    if ekf:
        b.enable_ekf()
        for f in fs:
            f.enable_ekf()
    for t in range(3):
        Phi = np.eye(n) + 1e-2 * rng.standard_normal((M, n, n))
        Ht = rng.standard_normal((M, p, n))
        real = rng.standard_normal((M, p)); comp = real + 1e-2 * rng.standard_normal((M, p))
        if t == 2:
            Ht[bad, 3] = 0.0
        b.prepare(Phi, Ht)
        est = b.update_nl(real, comp)
        for i, f in enumerate(fs):
            f.prepare(Phi[i], Ht[i])
            assert f.update_nl(real[i], comp[i]) == (orc.ERR_SINGULAR if (t == 2 and i in bad) else orc.OK)
        assert sorted(np.nonzero(b.status())[0].tolist()) == (bad if t == 2 else [])
        good = [i for i in range(M) if not (t == 2 and i in bad)]
        assert within(synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs])), 1e-9 if p <= n else 1e-7, "hybrid %d/%d state" % (n, p)), t
        assert synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs])) <= 1e-9, t
        assert synth.rel_frobenius(est.gain()[good], np.array([fs[i].gain() for i in good])) <= 1e-9, t
    assert [b.filter_step(i) for i in (bad[0], good[0])] == [2, 3]


@pytest.mark.parametrize("dtype,tol", [(k.F64, 1e-9), (k.F32, 2e-5)])
def test_every_shape_srif(dtype, tol):
    bad = []
    for n, p in GRID:
        rng = np.random.default_rng(700 * n + p)
        x0 = rng.standard_normal((N, n))
        P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
        R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))
        Phi, Ht, real, comp = _nl(n, p, rng)
        b = ga.FilterBatch(k.SRIF, n, p, 0, N, dtype=dtype, flags=k.FLAG_FULL_ESTIMATE if (n + p) % 2 else 0)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
        fs = [orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
        for t in range(STEPS):
            b.prepare(Phi[t], Ht[t])
            b.predict_nl() if t == 1 else b.update_nl(real[t], comp[t])
            for i, f in enumerate(fs):
                f.prepare(Phi[t, i], Ht[t, i])
                assert (f.predict_nl() if t == 1 else f.update_nl(real[t, i], comp[t, i])) == orc.OK
        eR = synth.rel_frobenius(b.get(k.RAW_MAT), np.array([f.raw_mat() for f in fs]))
        eb = synth.rel_frobenius(b.get(k.RAW_VEC), np.array([f.raw_vec() for f in fs]))
        if b.status().any() or b.step() != STEPS or max(eR, eb) > tol:
            bad.append((n, p, eR, eb))
    assert not bad, bad


@pytest.mark.parametrize("kind,okind", [(k.VANILLA, orc.VANILLA), (k.SQUAREROOT, orc.SQUAREROOT)])
@pytest.mark.parametrize("n,p,ntiles_extra", [(16, 4, 37), (14, 7, 8), (13, 2, 9), (9, 3, 23), (12, 8, 15), (7, 3, 5)])
def test_many_tiles_odd_counts(kind, okind, n, p, ntiles_extra):
    """Tile counts that are no multiple of 8 (the eight-lane kernels hand neighbouring parts to workgroups b and b + 8: the map from
    workgroup to part has to stay a bijection, kb_vanilla_split.h split_part_of_block) and a batch that ends inside a part."""
    Nb, steps = 64 * ntiles_extra + 11, 2
    d = _model(n, p, 0, 900 * n + p, N=Nb, STEPS=steps)
    b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    for t in range(steps):
        b.update(d["y"][t], snapshot=False)
    xo, Po, nerr = orc.ldkf_batch(okind, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
    assert nerr == 0 and not b.status().any() and b.step() == steps
    assert synth.rel_frobenius(b.get(k.STATE), xo) <= 1e-9
    assert synth.rel_frobenius(b.get(k.COVAR), Po) <= 1e-9


@pytest.mark.parametrize("kind", [k.VANILLA, k.SQUAREROOT, k.INFORMATION])
def test_every_shape_device_measurements_equal_host_measurements(kind):
    """kb_update_dev (planar device measurements and controls, leading dimension beyond N) == kb_update (host arrays), bit for bit, at
    every shape: the two entry points differ in where y and u come from, never in the kernel."""
    import torch
    bad = []
    ld = N + 13
    for n, p in GRID:
        m = (n + p) % 3
        d = _model(n, p, m, 1100 * n + p)
        flags = k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0
        h = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=flags)
        g = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=flags)
        yd = torch.zeros(STEPS, p, ld, dtype=torch.float64, device="cuda")
        yd[:, :, :N] = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
        ud = None
        if m:
            ud = torch.zeros(STEPS, m, ld, dtype=torch.float64, device="cuda")
            ud[:, :, :N] = torch.from_numpy(np.ascontiguousarray(d["u"].transpose(0, 2, 1))).cuda()
        torch.cuda.synchronize()   # (fills and copies run on torch's stream; the handle's does not wait for it)
        for t in range(STEPS):
            h.update(d["y"][t], None if m == 0 else d["u"][t], snapshot=False)
            g.update_dev(yd[t].data_ptr(), ld, ud[t].data_ptr() if m else None, ld if m else 0)
        g.synchronize()
        same = all(np.array_equal(h.get(f), g.get(f)) for f in ((k.RAW_VEC, k.RAW_MAT) if kind == k.INFORMATION else (k.STATE, k.COVAR)))
        if not same or g.status().any() or g.step() != STEPS:
            bad.append((n, p, m))
    assert not bad, bad


@pytest.mark.parametrize("kind", [k.VANILLA, k.SQUAREROOT, k.INFORMATION])
def test_every_shape_one_model_for_all_filters_equals_per_filter_copies(kind):
    """A batch whose model fields were all uploaded once (broadcast) == a batch given N copies of that model, bit for bit, at every shape
    (n <= 8: SHARED instantiations with scalar model loads; beyond: the split kernels read tile 0's block)."""
    bad = []
    flags = k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0
    tile = lambda M: None if M is None else np.broadcast_to(M, (N,) + M.shape).copy()
    for n, p in GRID:
        m = (n + p) % 3
        d = _model(n, p, m, 1300 * n + p)
        F, G, H, Q, R = d["F"][0], None if m == 0 else d["G"][0], d["H"][0], d["Q"][0], d["R"][0]
        shared = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], F, G, H, Q, R, nfilters=N, flags=flags)
        perf = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], tile(F), tile(G), tile(H), tile(Q), tile(R), flags=flags)
        for t in range(STEPS):
            shared.update(d["y"][t], None if m == 0 else d["u"][t], snapshot=False)
            perf.update(d["y"][t], None if m == 0 else d["u"][t], snapshot=False)
        same = all(np.array_equal(shared.get(f).view(np.uint64), perf.get(f).view(np.uint64)) for f in (k.RAW_VEC, k.RAW_MAT))
        if not same or shared.status().any() or not np.array_equal(shared.status(), perf.status()):
            bad.append((n, p, m))
    assert not bad, bad


@pytest.mark.parametrize("kind,okind", [(k.VANILLA, orc.VANILLA), (k.SQUAREROOT, orc.SQUAREROOT)])
def test_every_shape_awgn_replayed_through_the_oracle(kind, okind):
    """AWGN (noise.go:109-164) at every shape: the device's draws (kb_noise_sample) replayed through the oracle in the reference's call
    order (vanilla.go:146,157,195; squareroot.go draws Measurement(k) only into yhat)."""
    bad = []
    for n, p in GRID:
        m, full = (n + p) % 3, bool((n + p) % 2)
        d = _model(n, p, m, 1700 * n + p)
        b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], d["G"], d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE if full else 0,
                                    noise=k.NOISE_AWGN, seed=77)
        for t in range(STEPS):
            est = b.update(d["y"][t], d["u"][t] if m else None, snapshot=(t == STEPS - 1))
        check = list(range(0, N, 9)) + [N - 1]
        xs, Ps, ys = [], [], []
        for i in check:
            LQ, LR = orc.cholesky_lower(d["Q"][i])[1], orc.cholesky_lower(d["R"][i])[1]
            f = orc.Filter.ldkf(okind, d["x0"][i], d["P0"][i], d["F"][i], d["G"][i] if m else None, d["H"][i], d["Q"][i], d["R"][i])
            for t in range(STEPS):
                w0, v, w2 = LQ @ b.noise_sample(i, 0, t, 0, n), LR @ b.noise_sample(i, 0, t, 1, p), LQ @ b.noise_sample(i, 0, t, 2, n)
                assert f.update(d["y"][t, i], d["u"][t, i] if m else None, w_pred=w0, v_meas=v, w_post=w2) == orc.OK
            xs.append(f.state()); Ps.append(f.covariance()); ys.append(f.measurement())
        idx = np.array(check)
        ex = synth.rel_frobenius(est.state()[idx], np.array(xs))
        eP = synth.rel_frobenius(est.covariance()[idx], np.array(Ps))
        ey = synth.rel_frobenius(est.measurement()[idx], np.array(ys)) if full else 0.0
        if b.status().any() or b.step() != STEPS or max(ex, eP, ey) > 1e-9:
            bad.append((n, p, m, full, ex, eP, ey))
    assert not bad, bad


def test_every_shape_monte_carlo_and_chisquare_vs_oracle_replay():
    """NewMonteCarloRuns (montecarlo.go:92-119) and NewChiSquare (chisquare.go:16-95) at every shape: the runs replayed through the
    oracle with the device's draws, the NIS / NEES means against the oracle's restatement over those runs."""
    bad = []
    runs, steps = 70, 4
    for n, p in GRID:
        m = (n + p) % 3
        rng = np.random.default_rng(1900 * n + p)
        F = np.eye(n) + 0.05 * rng.standard_normal((n, n)); G = 0.3 * rng.standard_normal((n, m)) if m else None; H = rng.standard_normal((p, n))
        A = 0.1 * rng.standard_normal((n, n)); Q = A @ A.T + 1e-3 * np.eye(n)
        B = 0.2 * rng.standard_normal((p, p)); R = B @ B.T + 1e-2 * np.eye(p)
        x0, P0, mc_x0 = np.zeros(n), 1.5 * np.eye(n), 0.2 * rng.standard_normal(n)
        controls = rng.standard_normal((steps, m)) if m else np.zeros((1, 1))
        truth = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, mc_x0, P0, F, G, H, Q, R, nfilters=runs, noise=k.NOISE_AWGN, seed=9)
        kf = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, G, H, Q, R, nfilters=runs)
        mc = ga.new_monte_carlo_runs(runs, steps, p, controls, truth)
        nis, nees = ga.new_chi_square(kf, mc, controls)
        LQ, LR = orc.cholesky_lower(Q)[1], orc.cholesky_lower(R)[1]
        ts, tm = np.zeros((runs, steps, n)), np.zeros((runs, steps, p))
        for r in range(runs):
            f = orc.Filter.ldkf(orc.VANILLA_PREDICT, mc_x0, P0, F, G, H, Q, R)
            for t in range(steps):
                w = LQ @ truth.noise_sample(r, 0, t, 0, n); v = LR @ truth.noise_sample(r, 0, t, 1, p)
                assert f.update(np.zeros(p), controls[t] if m else None, w_pred=w, v_meas=v) == orc.OK
                ts[r, t], tm[r, t] = f.state(), f.measurement()

        def factory():
            f = orc.Filter.ldkf(orc.VANILLA, x0, P0, F, G, H, Q, R)
            f._H, f._R = H, R
            return f

        onis, onees = orc.chisquare(factory, ts, tm, controls if m else None)
        es, em = synth.rel_frobenius(mc._states(), ts), synth.rel_frobenius(mc._measurements(), tm)
        ok = es <= 1e-12 and em <= 1e-12 and np.allclose(nis, onis, rtol=1e-8) and np.allclose(nees, onees, rtol=1e-8)
        ok = ok and np.allclose(mc.mean(steps - 1), ts[:, steps - 1].mean(axis=0), rtol=1e-9, atol=1e-12)
        ok = ok and np.allclose(mc.stddev(steps - 1), ts[:, steps - 1].std(axis=0, ddof=1), rtol=1e-9, atol=1e-12)
        if not ok:
            bad.append((n, p, m, es, em))
    assert not bad, bad
