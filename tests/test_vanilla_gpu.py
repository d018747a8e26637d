"""Parity of the HIP Vanilla path (through the C ABI) against the CPU oracle and the
reference's jerkcar fixture.  Tolerance: 1e-9 relative Frobenius on state and covariance
(BASELINE.json north_star), 5.1e-7 abs against the %f-printed CSV."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth
from oracle import oracle as orc
from tests import jerkcar as jc
from tests.achieved import within

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _oracle_steps(d, kind, steps, want_extras=False):
    N = d["x0"].shape[0]
    xs, Ps, ex = [], [], []
    for i in range(N):
        f = orc.Filter.ldkf(kind, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
        for t in range(steps):
            assert f.update(d["y"][t, i]) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance())
        if want_extras:
            ex.append((f.pred_covariance(), f.gain(), f.innovation(), f.measurement()))
    return np.array(xs), np.array(Ps), ex


@pytest.mark.parametrize("N", [1, 64, 100, 4096])
def test_vanilla_6x3_host_path_vs_oracle(N):
    steps = 20
    d = synth.linear_batch(N, 6, 3, steps)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    for t in range(steps):
        b.update(d["y"][t])
    Nc = min(N, 256)
    sub = {kk: (v[:Nc] if kk != "y" else v[:, :Nc]) for kk, v in d.items()}
    xo, Po, _ = _oracle_steps(sub, orc.VANILLA, steps)
    assert synth.rel_frobenius(b.get(k.STATE, 0, Nc), xo) <= TOL
    assert synth.rel_frobenius(b.get(k.COVAR, 0, Nc), Po) <= TOL
    assert b.step() == steps
    assert not b.status().any()


def test_vanilla_6x3_full_estimate_extras():
    N, steps = 128, 5
    d = synth.linear_batch(N, 6, 3, steps)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"],
                                flags=k.FLAG_FULL_ESTIMATE)
    for t in range(steps):
        est = b.update(d["y"][t])
    xo, Po, ex = _oracle_steps(d, orc.VANILLA, steps, want_extras=True)
    assert synth.rel_frobenius(est.state(), xo) <= TOL
    assert synth.rel_frobenius(est.covariance(), Po) <= TOL
    assert synth.rel_frobenius(est.pred_covariance(), np.array([e[0] for e in ex])) <= TOL
    assert synth.rel_frobenius(est.gain(), np.array([e[1] for e in ex])) <= TOL
    assert np.max(np.abs(est.innovation() - np.array([e[2] for e in ex]))) <= 1e-9
    assert synth.rel_frobenius(est.measurement(), np.array([e[3] for e in ex])) <= TOL


def test_vanilla_generic_shapes_vs_oracle():
    for (n, p) in [(2, 1), (4, 1), (4, 2), (8, 3), (12, 6)]:
        N, steps = 70, 6
        d = synth.linear_batch(N, n, p, steps)
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"],
                                    flags=k.FLAG_FULL_ESTIMATE | k.FLAG_STRICT_SYMCHECK)
        for t in range(steps):
            est = b.update(d["y"][t])
        xo, Po, ex = _oracle_steps(d, orc.VANILLA, steps, want_extras=True)
        assert synth.rel_frobenius(est.state(), xo) <= TOL, (n, p)
        assert synth.rel_frobenius(est.covariance(), Po) <= TOL, (n, p)
        assert synth.rel_frobenius(est.gain(), np.array([e[1] for e in ex])) <= TOL, (n, p)
        assert not b.status().any()


def test_vanilla_jerkcar_fixture_on_gpu():
    """The reference's 2000-step golden run, H/noise swapped every 10th step, control input."""
    b = ga.FilterBatch.new_ldkf(k.VANILLA, jc.X0, jc.P0, jc.F, jc.G, jc.H2, jc.Q, jc.R2, nfilters=3, pmax=2)
    assert b.need_ctrl()

    def row():
        return jc.export_row(b.get(k.STATE, 1, 1)[0], b.get(k.COVAR, 1, 1)[0])

    got = jc.run_protocol(lambda y, u: b.update(y, u), b.set_measurement_matrix, b.set_noise, row)
    exp = jc.load_expected("vanilla")
    assert np.max(np.abs(got - exp)) <= 5.1e-7
    f = orc.Filter.ldkf(orc.VANILLA, jc.X0, jc.P0, jc.F, jc.G, jc.H2, jc.Q, jc.R2)
    ref = jc.run_protocol(lambda y, u: f.update(y, u), f.set_measurement_matrix, f.set_noise,
                          lambda: jc.export_row(f.state(), f.covariance()))
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) <= 1e-9


def test_vanilla_device_path_and_fused_steps():
    import torch
    N, steps = 4096 + 37, 12
    d = synth.linear_batch(N, 6, 3, steps)
    xo, Po, _ = _oracle_steps({kk: (v[:128] if kk != "y" else v[:, :128]) for kk, v in d.items()}, orc.VANILLA, steps)
    y_planar = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()  # [T][p][N]
    for fused in (False, True):
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        if fused:
            b.update_steps_dev(y_planar.data_ptr(), N, steps)
        else:
            for t in range(steps):
                b.update_dev(y_planar[t].data_ptr(), N)
        b.synchronize()
        assert b.step() == steps
        assert synth.rel_frobenius(b.get(k.STATE, 0, 128), xo) <= TOL
        assert synth.rel_frobenius(b.get(k.COVAR, 0, 128), Po) <= TOL


def test_fused_steps_on_ill_conditioned_innovation_covariance():
    """kb_update_steps_dev's time-fused kernel divides by Newton-refined reciprocals and evaluates the Joseph form in the distributed
    order (include/gokalman_amd.h): not the per-step kernel's bits.  Nearly collinear measurement rows and a small R make
    S = H P- H^T + R ill-conditioned (cond ~1e6..1e7); any backward-stable evaluation is then cond(S) x eps away from the oracle, and
    both device paths are held to 64 x cond(S) x 2^-53 -- printed next to what they achieve."""
    import torch
    N, steps = 2048, 20
    d = synth.linear_batch(N, 6, 3, steps, seed=synth.SEED + 5)
    rng = np.random.default_rng(55)
    d["H"][:, 1] = d["H"][:, 0] + 1e-3 * rng.standard_normal((N, 6))
    d["R"] = np.ascontiguousarray(np.broadcast_to(1e-7 * np.eye(3), (N, 3, 3)))
    M = 256
    fs = [orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i]) for i in range(M)]
    cond = 1.0
    for t in range(steps):
        for i, f in enumerate(fs):
            assert f.update(d["y"][t, i]) == orc.OK
            Pm = f.pred_covariance()
            cond = max(cond, np.linalg.cond(d["H"][i] @ Pm @ d["H"][i].T + d["R"][i]))
    xo, Po = np.array([f.state() for f in fs]), np.array([f.covariance() for f in fs])
    tol = 64 * cond * 2.0 ** -53
    y_planar = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()
    got = {}
    for fused in (False, True):
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        if fused:
            b.update_steps_dev(y_planar.data_ptr(), N, steps)
        else:
            for t in range(steps):
                b.update_dev(y_planar[t].data_ptr(), N)
        b.synchronize()
        assert b.step() == steps and not b.status().any()
        got[fused] = (synth.rel_frobenius(b.get(k.STATE, 0, M), xo), synth.rel_frobenius(b.get(k.COVAR, 0, M), Po))
    print("ill-conditioned S (max cond %.2e, bound %.2e): per-step kernel x %.2e P %.2e; fused kernel x %.2e P %.2e"
          % (cond, tol, got[False][0], got[False][1], got[True][0], got[True][1]))
    assert cond >= 1e5
    assert max(got[False]) <= tol and max(got[True]) <= tol


def test_vanilla_singular_innovation_sets_status_and_keeps_estimate():
    """H = 0 and R = 0 make H P- H^T + R exactly singular: the reference returns (nil, err)
    (vanilla.go:164-167) and leaves prevEst alone."""
    N = 70
    d = synth.linear_batch(N, 6, 3, 1)
    d["H"][5] = 0.0
    d["R"][5] = 0.0
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    b.update(d["y"][0])
    st = b.status()
    assert st[5] & k.ST_SINGULAR and not st[np.arange(N) != 5].any()
    assert np.array_equal(b.get(k.STATE, 5, 1)[0], d["x0"][5])
    f = orc.Filter.ldkf(orc.VANILLA, d["x0"][5], d["P0"][5], d["F"][5], None, d["H"][5], d["Q"][5], d["R"][5])
    assert f.update(d["y"][0, 5]) == orc.ERR_SINGULAR
    b.reset()
    assert not b.status().any() and b.step() == 0


def test_vanilla_dimension_errors_match_reference_strings():
    d = synth.linear_batch(4, 4, 2, 1)
    G = np.array([[0.0], [1e-4], [1e-2], [0.0]])
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], G, d["H"], d["Q"], d["R"])
    with pytest.raises(ga.KalmanError, match=r"dimensions must agree: measurement \(y\)\(3x\.\.\.\) H\(2x\.\.\.\)"):
        b.update(np.zeros((4, 3)), np.zeros((4, 1)))
    with pytest.raises(ga.KalmanError, match=r"dimensions must agree: control \(u\)\(2x\.\.\.\) G\(\.\.\.x1\)"):
        b.update(np.zeros((4, 2)), np.zeros((4, 2)))
    assert b.step() == 0


def test_vanilla_batch_noise_replay():
    """BatchNoise (noise.go:67-106): recorded process / measurement vectors, zero Q and R matrices."""
    N, n, p, steps = 70, 4, 2, 6
    d = synth.linear_batch(N, n, p, steps)
    rng = np.random.default_rng(9)
    proc, meas = 1e-2 * rng.standard_normal((steps, n)), 1e-2 * rng.standard_normal((steps, p))
    Z_Q, Z_R = np.zeros((n, n)), np.zeros((p, p))
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], Z_Q, Z_R, nfilters=N, flags=k.FLAG_FULL_ESTIMATE)
    b.set_batch_noise(proc, meas)
    for t in range(steps):
        est = b.update(d["y"][t])
    xs, ys = [], []
    for i in range(N):
        f = orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], Z_Q, Z_R)
        for t in range(steps):
            assert f.update(d["y"][t, i], None, proc[t], meas[t], proc[t]) == orc.OK
        xs.append(f.state()); ys.append(f.measurement())
    assert synth.rel_frobenius(est.state(), np.array(xs)) <= 1e-9
    assert synth.rel_frobenius(est.measurement(), np.array(ys)) <= 1e-9
    with pytest.raises(ga.KalmanError, match=r"no process noise defined at step k=6"):
        b.update(d["y"][0])


def _random_model(rng, N, n, p, m, steps):
    F = np.eye(n) + 0.05 * rng.standard_normal((N, n, n))
    H = rng.standard_normal((N, p, n))
    A = rng.standard_normal((N, n, n)); Q = 1e-3 * (A @ np.swapaxes(A, 1, 2)) + 1e-4 * np.eye(n)
    B = rng.standard_normal((N, p, p)); R = 1e-2 * (B @ np.swapaxes(B, 1, 2)) + 1e-2 * np.eye(p)
    G = rng.standard_normal((N, n, m)) if m else None
    x0 = rng.standard_normal((N, n)); P0 = np.tile(np.eye(n) * 2.0, (N, 1, 1))
    y = rng.standard_normal((steps, N, p)); u = rng.standard_normal((steps, N, m)) if m else None
    return F, G, H, Q, R, x0, P0, y, u


@pytest.mark.parametrize("n,p,m", [(1, 1, 0), (3, 2, 0), (5, 2, 0), (5, 3, 0), (5, 4, 0), (7, 1, 0), (7, 3, 0), (8, 4, 0), (6, 4, 0),
                                   (3, 1, 2), (5, 3, 1), (8, 4, 2), (2, 2, 1)])
@pytest.mark.parametrize("full", [False, True])
def test_vanilla_padded_register_kernels_vs_oracle(n, p, m, full):
    """Shapes without an exact register kernel run on the next larger padded instantiation (kb_vanilla_pad.hip):
    results must be those of the oracle at the real dimensions, extras included."""
    rng = np.random.default_rng(1000 * n + 10 * p + m)
    N, steps = 130, 5
    F, G, H, Q, R, x0, P0, y, u = _random_model(rng, N, n, p, m, steps)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, G, H, Q, R, flags=k.FLAG_FULL_ESTIMATE if full else 0)
    for t in range(steps):
        est = b.update(y[t], u[t] if m else None)
    xs, Ps, Ks, Pm, inn = [], [], [], [], []
    for i in range(N):
        f = orc.Filter.ldkf(orc.VANILLA, x0[i], P0[i], F[i], G[i] if m else None, H[i], Q[i], R[i])
        for t in range(steps):
            assert f.update(y[t, i], u[t, i] if m else None) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance()); Ks.append(f.gain()); Pm.append(f.pred_covariance()); inn.append(f.innovation())
    assert synth.rel_frobenius(est.state(), np.array(xs)) <= TOL
    assert synth.rel_frobenius(est.covariance(), np.array(Ps)) <= TOL
    if full:
        assert synth.rel_frobenius(est.gain(), np.array(Ks)) <= TOL
        assert synth.rel_frobenius(est.pred_covariance(), np.array(Pm)) <= TOL
        assert synth.rel_frobenius(est.innovation(), np.array(inn)) <= TOL
    assert not b.status().any()


@pytest.mark.parametrize("kind,okind", [(k.VANILLA, orc.VANILLA), (k.SQUAREROOT, orc.SQUAREROOT)])
def test_parity_gate_4096_filters_100_steps(kind, okind):
    """SURVEY 8d's parity gate for configs B and C as written: the first 4096 filters x 100 steps of the benchmark batch
    (same generator, same seed), device-resident measurements, max relative Frobenius error <= 1e-9 for x and P."""
    import torch
    N, T, n, p = 4096, 100, 6, 3
    d = synth.linear_batch(N, n, p, T)
    b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0, 2, 1))).cuda()      # planar [T][p][N]
    for t in range(T):
        b.update_dev(y[t].data_ptr(), N)
    b.synchronize()
    xo, Po, nerr = orc.ldkf_batch(okind, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
    assert nerr == 0 and not b.status().any()
    x, P = b.get(k.STATE), b.get(k.COVAR)
    ex = np.linalg.norm(x - xo, axis=1) / np.linalg.norm(xo, axis=1)
    eP = np.linalg.norm((P - Po).reshape(N, -1), axis=1) / np.linalg.norm(Po.reshape(N, -1), axis=1)
    assert ex.max() <= 1e-9 and eP.max() <= 1e-9, (ex.max(), eP.max())


@pytest.mark.parametrize("n,p,m,full,predict", [(6, 3, 0, False, False), (6, 3, 0, True, False), (4, 2, 0, False, False), (4, 2, 2, True, False),
                                                (2, 1, 1, True, False), (5, 3, 1, False, False), (8, 4, 2, True, False), (7, 2, 0, False, False),
                                                (6, 3, 0, True, True), (4, 2, 2, False, True)])
def test_vanilla_awgn_on_the_register_kernels_replayed_through_the_oracle(n, p, m, full, predict):
    """AWGN (noise.go:109-164) on the register kernels (kb_vanilla_reg.h, NOISE): the device's draws (kb_noise_sample: the
    standard normals of (filter, epoch, kf.step, which)) are replayed through the oracle in the reference's call order --
    Process(k) into x-, Measurement(k) into yhat, Process(k) again into x+ (vanilla.go:146,157,195) -- 4096 filters x 20
    steps for the benchmark shape, every other register-kernel family member (exact, padded, with / without control, pure
    predictor) on a smaller batch."""
    bench_shape = (n, p, m) == (6, 3, 0) and not predict
    N, steps = (4096, 20) if bench_shape else (192, 6)
    rng = np.random.default_rng(100 * n + 10 * p + m + (7 if full else 0))
    F, G, H, Q, R, x0, P0, y, u = _random_model(rng, N, n, p, m, steps)
    kind, okind = (k.VANILLA_PREDICT, orc.VANILLA_PREDICT) if predict else (k.VANILLA, orc.VANILLA)
    b = ga.FilterBatch.new_ldkf(kind, x0, P0, F, G, H, Q, R, flags=k.FLAG_FULL_ESTIMATE if full else 0, noise=k.NOISE_AWGN, seed=4711)
    for t in range(steps):
        est = b.update(y[t], u[t] if m else None, snapshot=(t == steps - 1))
    check = range(N) if not bench_shape else list(range(0, N, 37)) + [N - 1]
    xs, Ps, ys, inn = [], [], [], []
    for i in check:
        LQ, LR = orc.cholesky_lower(Q[i])[1], orc.cholesky_lower(R[i])[1]
        f = orc.Filter.ldkf(okind, x0[i], P0[i], F[i], G[i] if m else None, H[i], Q[i], R[i])
        for t in range(steps):
            w0, v, w2 = LQ @ b.noise_sample(i, 0, t, 0, n), LR @ b.noise_sample(i, 0, t, 1, p), LQ @ b.noise_sample(i, 0, t, 2, n)
            assert f.update(y[t, i], u[t, i] if m else None, w_pred=w0, v_meas=v, w_post=w2) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance()); ys.append(f.measurement()); inn.append(f.innovation())
    idx = np.array(list(check))
    assert synth.rel_frobenius(est.state()[idx], np.array(xs)) <= TOL
    assert synth.rel_frobenius(est.covariance()[idx], np.array(Ps)) <= TOL
    if full:
        assert synth.rel_frobenius(est.measurement()[idx], np.array(ys)) <= TOL
        if not predict:
            assert within(synth.rel_frobenius(est.innovation()[idx], np.array(inn)), 1e-7)   # differences of O(1) numbers: absolute 1e-16
    assert not b.status().any() and b.step() == steps


def test_vanilla_noise_register_kernel_is_the_one_that_runs():
    """The AWGN batch above must not fall back to the scratch-array generic kernel (the 16-30x cliff of round 2): the register
    kernel and the generic kernel differ in FMA contraction, so their results differ in the last bits, while a second register
    run is bit-identical.  KB_FLAG_STRICT_SYMCHECK forces the generic kernel."""
    N, n, p, steps = 256, 6, 3, 5
    d = synth.linear_batch(N, n, p, steps)
    runs = []
    for flags in (0, 0, k.FLAG_STRICT_SYMCHECK):
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=flags, noise=k.NOISE_AWGN, seed=9)
        for t in range(steps):
            b.update(d["y"][t])
        runs.append((b.get(k.STATE), b.get(k.COVAR)))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert not np.array_equal(runs[0][1], runs[2][1])                      # another kernel ...
    assert synth.rel_frobenius(runs[0][0], runs[2][0]) <= 1e-12 and synth.rel_frobenius(runs[0][1], runs[2][1]) <= 1e-12   # ... same filter


def test_device_normals_are_bit_identical_to_the_host_replay():
    """The normals a kernel draws and the ones kb_noise_sample replays on the host come from the same arithmetic
    (csrc/kb_normal.h: explicit FMAs, correctly rounded division and sqrt): with F = 0, x0 = 0 and Q = I the pure predictor's
    x- IS the process draw of the step (vanilla.go:146 with chol(I) = I), so the device's bits can be read off and compared."""
    N, n, p, steps = 4096 + 5, 6, 3, 3
    Z = np.zeros
    H = np.tile(np.eye(p, n), (N, 1, 1))
    b = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, Z((N, n)), np.tile(np.eye(n), (N, 1, 1)), Z((N, n, n)), None, H,
                                np.tile(np.eye(n), (N, 1, 1)), np.tile(np.eye(p), (N, 1, 1)), noise=k.NOISE_AWGN, seed=20161103)
    for t in range(steps):
        b.update(Z((N, p)), snapshot=False)
        X = b.get(k.STATE)
        for i in list(range(0, N, 97)) + [N - 1]:
            want = b.noise_sample(i, 0, t, 0, n)
            assert np.array_equal(X[i].view(np.uint64), want.view(np.uint64)), (t, i, X[i], want)
    # and they are standard normals: mean 0, variance 1 over 6 x 4101 draws of the last step
    assert abs(X.mean()) < 0.03 and abs(X.var() - 1.0) < 0.05 and np.abs(X).max() < 6.5


@pytest.mark.parametrize("n,p,m,noise,full", [(6, 3, 0, k.NOISE_NOISELESS, False), (6, 3, 0, k.NOISE_AWGN, True), (5, 2, 1, k.NOISE_NOISELESS, True),
                                              (8, 4, 2, k.NOISE_NOISELESS, False), (4, 2, 0, k.NOISE_NOISELESS, False), (3, 1, 0, k.NOISE_AWGN, False)])
def test_shared_model_batch_equals_the_per_filter_batch_bit_for_bit(n, p, m, noise, full):
    """A batch whose model fields were all uploaded with broadcast = 1 runs the SHARED instantiations (kb_vanilla_shared.hip: every
    wave reads tile 0's model block, default cache policy): the same arithmetic as the per-filter kernels, so the results equal
    those of a batch given N copies of that model, bit for bit.  Then ONE field becomes per-filter (SetStateTransition with N
    matrices): the batch leaves the shared path and both batches still agree; back to one F for all: shared again."""
    N, steps = 4096 + 37, 4
    rng = np.random.default_rng(7 * n + p + m)
    F, G, H, Q, R, x0, P0, y, u = _random_model(rng, N, n, p, m, 3 * steps)
    F1, H1, Q1, R1 = F[0], H[0], Q[0], R[0]
    G1 = G[0] if m else None
    flags = k.FLAG_FULL_ESTIMATE if full else 0
    tile = lambda M: np.broadcast_to(M, (N,) + M.shape).copy()
    shared = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F1, G1, H1, Q1, R1, nfilters=N, flags=flags, noise=noise, seed=3)
    perf = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, tile(F1), tile(G1) if m else None, tile(H1), tile(Q1), tile(R1), flags=flags, noise=noise, seed=3)

    def run(t0):
        for t in range(t0, t0 + steps):
            for b in (shared, perf):
                b.update(y[t], u[t] if m else None, snapshot=False)
        for f in [k.STATE, k.COVAR] + ([k.PRED_COVAR, k.GAIN, k.INNOVATION, k.MEASUREMENT] if full else []):
            assert np.array_equal(shared.get(f).view(np.uint64), perf.get(f).view(np.uint64)), f
        assert not shared.status().any() and not perf.status().any()
    run(0)
    shared.set(k.F, F, 2); perf.set(k.F, F, 2)            # N different transition matrices: no longer one model
    run(steps)
    if noise == k.NOISE_NOISELESS and not m:
        f = orc.Filter.ldkf(orc.VANILLA, x0[5], P0[5], F1, None, H1, Q1, R1)
        for t in range(2 * steps):
            if t == steps:
                f.set_state_transition(F[5])
            assert f.update(y[t, 5]) == orc.OK
        assert synth.rel_frobenius(shared.get(k.STATE, 5, 1)[0], f.state()) <= TOL and synth.rel_frobenius(shared.get(k.COVAR, 5, 1)[0], f.covariance()) <= TOL
    shared.set(k.F, F1, 2); perf.set(k.F, tile(F1), 2)    # one F for all again
    run(2 * steps)
