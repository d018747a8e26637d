"""Edge cases through the C ABI: maximum dimensions, fp32 batches, partial reads, reset / re-run
determinism, setter side effects (reference behaviours cited per test)."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth
from oracle import oracle as orc
from tests.achieved import within

pytestmark = pytest.mark.gpu


def test_maximum_dimensions_16x8_generic_kernel():
    N, n, p, steps = 33, 16, 8, 3
    d = synth.linear_batch(N, n, p, steps)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE)
    for t in range(steps):
        est = b.update(d["y"][t])
    xs, Ps = [], []
    for i in range(N):
        f = orc.Filter.ldkf(orc.VANILLA, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
        for t in range(steps):
            assert f.update(d["y"][t, i]) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance())
    assert synth.rel_frobenius(est.state(), np.array(xs)) <= 1e-9
    assert synth.rel_frobenius(est.covariance(), np.array(Ps)) <= 1e-9
    with pytest.raises(ga.KalmanError):
        ga.FilterBatch(k.VANILLA, 17, 3, 0, 4)


@pytest.mark.parametrize("kind,okind", [(k.VANILLA, orc.VANILLA), (k.SQUAREROOT, orc.SQUAREROOT)])
def test_fp32_batches_track_the_fp64_oracle(kind, okind):
    """fp32 storage + arithmetic (the SRIF config's dtype) on the LDKF kinds: fp32-appropriate tolerance."""
    N, steps = 256, 10
    d = synth.linear_batch(N, 6, 3, steps)
    b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], dtype=k.F32)
    for t in range(steps):
        b.update(d["y"][t])
    xo, Po, nerr = orc.ldkf_batch(okind, d["x0"], d["P0"], d["F"], d["H"], d["Q"], d["R"], d["y"])
    assert nerr == 0
    assert synth.rel_frobenius(b.get(k.STATE), xo) <= 5e-3
    assert synth.rel_frobenius(b.get(k.COVAR), Po) <= 5e-3


def test_partial_reads_and_reset_rerun_is_bitwise_repeatable():
    N, steps = 1000, 5
    d = synth.linear_batch(N, 6, 3, steps)
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    for t in range(steps):
        b.update(d["y"][t])
    full = b.get(k.COVAR)
    assert np.array_equal(b.get(k.COVAR, 137, 64), full[137:201])
    assert np.array_equal(b.get(k.STATE, 999, 1), b.get(k.STATE)[999:])
    x1 = b.get(k.STATE)
    b.reset()                                    # vanilla.go:121-125
    assert b.step() == 0 and np.array_equal(b.get(k.STATE), d["x0"])
    for t in range(steps):
        b.update(d["y"][t])
    assert np.array_equal(b.get(k.STATE), x1) and np.array_equal(b.get(k.COVAR), full)
    with pytest.raises(ga.KalmanError):
        b.get(k.STATE, 990, 20)
    with pytest.raises(ga.KalmanError, match="FULL_ESTIMATE"):
        b.get(k.GAIN)


def test_set_state_transition_and_noise_between_steps():
    """Set* between steps (kalman.go:41-44); Information refreshes F^-1 but not Q^-1/R^-1
    (information.go:117-138), SquareRoot recomputes chol(Q), chol(R) (squareroot.go:100-114)."""
    N, n, p = 40, 4, 2
    d = synth.linear_batch(N, n, p, 4)
    d2 = synth.linear_batch(N, n, p, 1, seed=99)
    for kind, okind, tol in ((k.VANILLA, orc.VANILLA, 1e-9), (k.SQUAREROOT, orc.SQUAREROOT, 1e-9), (k.INFORMATION, orc.INFORMATION, 1e-6)):
        flags = k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0
        b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=flags)
        fs = []
        for i in range(N):
            if okind == orc.INFORMATION:
                fs.append(orc.Filter.information_from_state(d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i]))
            else:
                fs.append(orc.Filter.ldkf(okind, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i]))
        for t in range(4):
            if t == 2:
                b.set_state_transition(d2["F"]); b.set_noise(d2["Q"], d2["R"])
                for i, f in enumerate(fs):
                    f.set_state_transition(d2["F"][i]); f.set_noise(d2["Q"][i], d2["R"][i])
            b.update(d["y"][t])
            for i, f in enumerate(fs):
                assert f.update(d["y"][t, i]) == orc.OK
        assert within(synth.rel_frobenius(b.get(k.STATE), np.array([f.state() for f in fs])), tol, "state"), kind
        assert within(synth.rel_frobenius(b.get(k.COVAR), np.array([f.covariance() for f in fs])), tol, "covariance"), kind


def test_is_within_nsigma_matches_oracle():
    N, steps = 300, 4
    d = synth.linear_batch(N, 6, 3, steps)
    for kind, okind in ((k.VANILLA, orc.VANILLA), (k.SQUAREROOT, orc.SQUAREROOT)):
        b = ga.FilterBatch.new_ldkf(kind, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        fs = [orc.Filter.ldkf(okind, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i]) for i in range(N)]
        for t in range(steps):
            est = b.update(d["y"][t])
            for i, f in enumerate(fs):
                f.update(d["y"][t, i])
        mixed = False
        for ns in (2.0, 20.0, 100.0, 1000.0):
            got = est.is_within_nsigma(ns)
            exp = np.array([f.is_within_nsigma(ns) for f in fs])
            assert np.mean(got == exp) >= 0.995   # ties at the boundary may flip with rounding
            mixed = mixed or (0 < got.sum() < N)
        assert mixed


def test_predict_only_vanilla_matches_oracle():
    """NewPurePredictorVanilla (vanilla.go:43-62, :170-179): estimate = {x-, yhat, 0, P-, P-, K}."""
    N, steps = 128, 6
    d = synth.linear_batch(N, 6, 3, steps)
    b = ga.FilterBatch.new_ldkf(k.VANILLA_PREDICT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE)
    for t in range(steps):
        est = b.update(np.zeros((N, 3)))
    xs, Ps, Ks = [], [], []
    for i in range(N):
        f = orc.Filter.ldkf(orc.VANILLA_PREDICT, d["x0"][i], d["P0"][i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
        for t in range(steps):
            assert f.update(np.zeros(3)) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance()); Ks.append(f.gain())
    assert synth.rel_frobenius(est.state(), np.array(xs)) <= 1e-9
    assert synth.rel_frobenius(est.covariance(), np.array(Ps)) <= 1e-9
    assert synth.rel_frobenius(est.pred_covariance(), np.array(Ps)) <= 1e-9
    assert synth.rel_frobenius(est.gain(), np.array(Ks)) <= 1e-9
    assert np.all(est.innovation() == 0)


def test_device_side_set_and_get_planar_roundtrip():
    """kb_set_dev / kb_get_dev: planar device arrays in, planar device arrays out (no PCIe)."""
    import torch
    N, n, p, steps = 777, 6, 3, 3
    d = synth.linear_batch(N, n, p, steps)
    ref = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    b = ga.FilterBatch(k.VANILLA, n, p, 0, N)

    def planar(a):  # [N, ...] -> [elems, N] on the device
        return torch.from_numpy(np.ascontiguousarray(a.reshape(N, -1).T)).cuda()

    keep = []
    for field, arr, pr in ((k.X, d["x0"], 0), (k.P, d["P0"], 0), (k.F, d["F"], 0), (k.H, d["H"], p), (k.Q, d["Q"], 0), (k.R, d["R"], p)):
        t = planar(arr); keep.append(t)
        b.set_dev(field, t.data_ptr(), N, p_rows=pr)
    b.init()
    for t in range(steps):
        ref.update(d["y"][t]); b.update(d["y"][t])
    assert np.array_equal(b.get(k.STATE), ref.get(k.STATE)) and np.array_equal(b.get(k.COVAR), ref.get(k.COVAR))
    xs = torch.zeros(n, N + 5, dtype=torch.float64, device="cuda"); Ps = torch.zeros(n * n, N + 5, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()   # (the zero fills run on torch's stream; the handle's does not wait for it)
    k.check(k.lib().kb_get_dev(b._h, k.STATE, xs.data_ptr(), N + 5))
    k.check(k.lib().kb_get_dev(b._h, k.COVAR, Ps.data_ptr(), N + 5))
    b.synchronize()
    assert np.array_equal(xs.cpu().numpy()[:, :N].T, ref.get(k.STATE))
    assert np.array_equal(Ps.cpu().numpy()[:, :N].T.reshape(N, n, n), ref.get(k.COVAR))


def test_create_destroy_does_not_leak_device_memory():
    import torch
    d = synth.linear_batch(20000, 6, 3, 1)

    def cycle():
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"], flags=k.FLAG_FULL_ESTIMATE)
        b.update(d["y"][0]); b.get(k.COVAR); b.is_within_nsigma(2.0)
        s = ga.FilterBatch.new_ldkf(k.SQUAREROOT, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        s.update(d["y"][0]); s.get(k.COVAR)
        b.close(); s.close()

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, (free0, free1)


def test_large_batch_4m_filters_properties():
    """4M filters (beyond what the oracle can replay): size-independent properties -- every filter finite and
    status-clean, covariance diagonals positive, and shard-invariance: filter i of a 4M batch equals filter i
    of a 4096-filter batch built from the same rows."""
    N, small = 1 << 22, 4096
    rng = np.random.default_rng(0)
    base = synth.linear_batch(small, 6, 3, 2)
    rep = N // small
    big = {kk: (np.tile(v, (rep,) + (1,) * (v.ndim - 1)) if kk != "y" else np.tile(v, (1, rep, 1))) for kk, v in base.items()}
    b = ga.FilterBatch.new_ldkf(k.VANILLA, big["x0"], big["P0"], big["F"], None, big["H"], big["Q"], big["R"])
    s = ga.FilterBatch.new_ldkf(k.VANILLA, base["x0"], base["P0"], base["F"], None, base["H"], base["Q"], base["R"])
    for t in range(2):
        b.update(big["y"][t]); s.update(base["y"][t])
    assert not b.status().any()
    tail = b.get(k.COVAR, N - small, small)
    assert np.array_equal(tail, s.get(k.COVAR)) and np.array_equal(b.get(k.STATE, N - small, small), s.get(k.STATE))
    assert np.all(np.isfinite(tail)) and np.all(np.diagonal(tail, axis1=1, axis2=2) > 0)
    del rng


def test_nan_measurement_is_contained_to_its_filter():
    """A NaN measurement makes that filter's update non-finite: AsSymDense's comparison fails on NaN in the
    reference (helper.go:75, vanilla.go:207-215 -> (nil, err), estimate untouched); other filters are unaffected."""
    N = 130
    d = synth.linear_batch(N, 6, 3, 2)
    d["y"][0, 77, 1] = np.nan
    b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
    b.update(d["y"][0])
    st = b.status()
    assert st[77] & k.ST_NONFINITE and not st[np.arange(N) != 77].any()
    assert np.array_equal(b.get(k.STATE, 77, 1)[0], d["x0"][77])
    f = orc.Filter.ldkf(orc.VANILLA, d["x0"][77], d["P0"][77], d["F"][77], None, d["H"][77], d["Q"][77], d["R"][77])
    assert f.update(d["y"][0, 77]) != orc.OK or not np.all(np.isfinite(f.state()))
    ok = np.arange(N) != 77
    xo, Po, _ = orc.ldkf_batch(orc.VANILLA, d["x0"][ok], d["P0"][ok], d["F"][ok], d["H"][ok], d["Q"][ok], d["R"][ok], d["y"][:1, ok])
    assert synth.rel_frobenius(b.get(k.STATE)[ok], xo) <= 1e-9


def test_two_handles_driven_from_two_threads_concurrently():
    """Threading contract of the C ABI (include/gokalman_amd.h): a handle is single-threaded like a reference filter, distinct
    handles may be driven from distinct threads (one HIP stream each).  Two threads step two batches at the same time; each
    must end bit-identical to the same batch stepped alone."""
    import threading
    N, steps = 4096, 40
    data = [synth.linear_batch(N, 6, 3, steps, seed=synth.SEED + 10 + i) for i in range(2)]
    kinds = [k.VANILLA, k.SQUAREROOT]

    def run(i, out):
        d = data[i]
        b = ga.FilterBatch.new_ldkf(kinds[i], d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
        for t in range(steps):
            b.update(d["y"][t])
        out[i] = (b.get(k.STATE), b.get(k.COVAR), int(np.count_nonzero(b.status())))

    alone, together = {}, {}
    for i in range(2):
        run(i, alone)
    threads = [threading.Thread(target=run, args=(i, together)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        assert together[i][2] == 0
        assert np.array_equal(alone[i][0], together[i][0]) and np.array_equal(alone[i][1], together[i][1])


def test_many_handles_with_big_shape_generic_steps_do_not_exhaust_scratch():
    """The run-time-dimension kernels for n > 8 keep up to 31 KB of private arrays per lane; the runtime sizes a queue's
    scratch for the whole device (15 GiB) and keeps it, and a process that spread such launches over its handles' own
    streams ran out of scratch (the runtime aborts).  They all go through one stream per device now (kb_internal.h
    HeavyScope): the retained scratch stays at one queue's worth however many handles there are."""
    import torch
    n, p, N = 12, 4, 64
    rng = np.random.default_rng(12)
    free0 = torch.cuda.mem_get_info()[0]
    keep = []
    for kind in (k.VANILLA, k.SQUAREROOT, k.INFORMATION, k.VANILLA, k.SQUAREROOT, k.INFORMATION, k.VANILLA, k.VANILLA):
        F = np.eye(n) + 1e-2 * rng.standard_normal((N, n, n))
        H = rng.standard_normal((N, p, n))
        b = ga.FilterBatch.new_ldkf(kind, rng.standard_normal((N, n)), np.eye(n), F, None, H, 1e-3 * np.eye(n), 0.1 * np.eye(p),
                                    flags=k.FLAG_STRICT_SYMCHECK | (k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0))
        est = b.update(rng.standard_normal((N, p)))
        assert np.isfinite(est.state()).all() and not b.status().any()
        keep.append(b)
    A = -np.eye(8) + 0.1 * rng.standard_normal((8, 8))
    Fd, Qd, st = ga.van_loan(A, np.eye(8), 1e-2 * np.eye(8), 0.01)
    assert np.isfinite(Fd).all() and np.isfinite(Qd).all()
    torch.cuda.synchronize()
    held = (free0 - torch.cuda.mem_get_info()[0]) / 2 ** 30
    assert held < 24.0, "scratch retained by the process: %.1f GiB" % held    # one queue's worth (<= 15.3 GiB) + the batches


@pytest.mark.parametrize("kind,flags", [(k.SQUAREROOT, 0), (k.INFORMATION, k.FLAG_INFO_FROM_STATE)])
def test_baseline_size_1m_filters_properties_other_kinds(kind, flags):
    """Config C's size (1M filters) for SquareRoot and Information: beyond what the oracle replays, so size-independent
    properties -- status-clean, finite, positive covariance diagonals, and shard invariance: the last 4096 filters of the
    1M batch are bit-equal to a 4096-filter batch fed the same rows (the oracle-checked size, tests/test_kinds_gpu.py)."""
    N, small = 1 << 20, 4096
    base = synth.linear_batch(small, 6, 3, 3)
    rep = N // small
    big = {kk: (np.tile(v, (rep,) + (1,) * (v.ndim - 1)) if kk != "y" else np.tile(v, (1, rep, 1))) for kk, v in base.items()}
    b = ga.FilterBatch.new_ldkf(kind, big["x0"], big["P0"], big["F"], None, big["H"], big["Q"], big["R"], flags=flags)
    s = ga.FilterBatch.new_ldkf(kind, base["x0"], base["P0"], base["F"], None, base["H"], base["Q"], base["R"], flags=flags)
    for t in range(3):
        b.update(big["y"][t]); s.update(base["y"][t])
    assert not b.status().any()
    tail_P, tail_x = b.get(k.COVAR, N - small, small), b.get(k.STATE, N - small, small)
    assert np.array_equal(tail_P, s.get(k.COVAR)) and np.array_equal(tail_x, s.get(k.STATE))
    assert np.all(np.isfinite(tail_P)) and np.all(np.diagonal(tail_P, axis1=1, axis2=2) > 0)
    head_P = b.get(k.COVAR, 0, small)
    assert np.array_equal(head_P, tail_P)           # the same rows at the other end of the batch
    assert np.allclose(tail_P, np.swapaxes(tail_P, 1, 2), rtol=1e-12, atol=1e-300)


def test_baseline_size_1m_hybrid_ekf_properties():
    """Config D(ii)'s size: 1M Hybrid EKF filters with per-step Phi / Htilde read in place from planar device arrays; shard
    invariance against a 4096-filter batch given the same columns, status-clean, symmetric positive covariance."""
    import torch
    N, small, n, p = 1 << 20, 4096, 6, 2
    rng = np.random.default_rng(12)
    x0s = rng.standard_normal((small, n))
    P0 = np.diag([10.0, 10, 10, 1, 1, 1])
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    Phi_s = (torch.eye(n, dtype=torch.float64, device="cuda").reshape(n * n, 1) + 1e-2 * torch.randn(n * n, small, dtype=torch.float64, device="cuda", generator=gen)).contiguous()
    Ht_s = torch.randn(p * n, small, dtype=torch.float64, device="cuda", generator=gen)
    real_s = torch.randn(p, small, dtype=torch.float64, device="cuda", generator=gen)
    comp_s = real_s + 1e-3 * torch.randn(p, small, dtype=torch.float64, device="cuda", generator=gen)
    rep = N // small
    Phi, Ht, real, comp = (v.repeat(1, rep).contiguous() for v in (Phi_s, Ht_s, real_s, comp_s))
    torch.cuda.synchronize()   # the handles' streams do not wait for torch's

    def make(M, x0):
        h = ga.FilterBatch(k.HYBRID, n, p, 0, M)
        h.set(k.X, x0, 1); h.set(k.P, P0, 2); h.set(k.R, np.diag([1e-6, 1e-6]), 2, p_rows=p); h.init(); h.enable_ekf()
        return h
    big, ref = make(N, np.tile(x0s, (rep, 1))), make(small, x0s)
    for _ in range(3):
        for h, (F_, H_, r_, c_, M) in ((big, (Phi, Ht, real, comp, N)), (ref, (Phi_s, Ht_s, real_s, comp_s, small))):
            k.check(k.lib().kb_prepare_dev(h._h, F_.data_ptr(), H_.data_ptr(), M))
            k.check(k.lib().kb_update_nl_dev(h._h, r_.data_ptr(), c_.data_ptr(), M))
    assert not big.status().any()
    tail_P = big.get(k.COVAR, N - small, small)
    assert np.array_equal(tail_P, ref.get(k.COVAR)) and np.array_equal(big.get(k.STATE, N - small, small), ref.get(k.STATE))
    assert np.all(np.isfinite(tail_P)) and np.all(np.diagonal(tail_P, axis1=1, axis2=2) > 0)


def test_baseline_size_1m_vanilla_awgn_sampled_oracle_replay():
    """Config B's size with the reference's usual Noise object (AWGN): 1M filters on the NOISE register kernel, 3 steps.
    Beyond a full oracle replay, so (a) 200 filters spread over the batch (first / last tile, tile borders) are replayed
    through the oracle with the device's own draws (kb_noise_sample, keyed by the GLOBAL filter index), (b) every filter is
    status-clean and finite, (c) identical rows given different filter indices end in DIFFERENT states (independent noise
    streams) while the covariances -- which the noise vectors never enter (vanilla.go:152, :197-205) -- are bit-equal."""
    N, small, steps = 1 << 20, 4096, 3
    base = synth.linear_batch(small, 6, 3, steps)
    rep = N // small
    big = {kk: (np.tile(v, (rep,) + (1,) * (v.ndim - 1)) if kk != "y" else np.tile(v, (1, rep, 1))) for kk, v in base.items()}
    b = ga.FilterBatch.new_ldkf(k.VANILLA, big["x0"], big["P0"], big["F"], None, big["H"], big["Q"], big["R"], noise=k.NOISE_AWGN, seed=31)
    for t in range(steps):
        b.update(big["y"][t], snapshot=False)
    assert not b.status().any() and b.step() == steps
    idx = np.unique(np.concatenate([np.arange(0, 70), np.arange(N - 70, N), np.random.default_rng(5).integers(0, N, 60)]))
    X = np.concatenate([b.get(k.STATE, int(i), 1) for i in idx]); P = np.concatenate([b.get(k.COVAR, int(i), 1) for i in idx])
    xs, Ps = [], []
    for i in idx:
        r = int(i) % small
        LQ, LR = orc.cholesky_lower(base["Q"][r])[1], orc.cholesky_lower(base["R"][r])[1]
        f = orc.Filter.ldkf(orc.VANILLA, base["x0"][r], base["P0"][r], base["F"][r], None, base["H"][r], base["Q"][r], base["R"][r])
        for t in range(steps):
            draws = [b.noise_sample(int(i), 0, t, w, 6 if w != 1 else 3) for w in range(3)]
            assert f.update(base["y"][t, r], None, w_pred=LQ @ draws[0], v_meas=LR @ draws[1], w_post=LQ @ draws[2]) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance())
    assert synth.rel_frobenius(X, np.array(xs)) <= 1e-9 and synth.rel_frobenius(P, np.array(Ps)) <= 1e-9
    head_x, tail_x = b.get(k.STATE, 0, small), b.get(k.STATE, N - small, small)
    head_P, tail_P = b.get(k.COVAR, 0, small), b.get(k.COVAR, N - small, small)
    assert np.isfinite(tail_x).all() and np.isfinite(tail_P).all()
    assert np.array_equal(head_P, tail_P) and not np.any(np.all(head_x == tail_x, axis=1))
    # the noise has the right size: (x_noisy - x_noiseless) has per-component spread of the order of sqrt(Q_ii) after 3 steps
    s = ga.FilterBatch.new_ldkf(k.VANILLA, base["x0"], base["P0"], base["F"], None, base["H"], base["Q"], base["R"])
    for t in range(steps):
        s.update(base["y"][t], snapshot=False)
    dev = tail_x - s.get(k.STATE)
    ratio = dev.std(axis=0) / np.sqrt(np.diagonal(base["Q"], axis1=1, axis2=2).mean(axis=0))
    assert np.all(ratio > 0.05) and np.all(ratio < 20.0), ratio


def test_baseline_size_1m_vanilla_strict_symcheck_register_kernel():
    """Config B's size under KB_FLAG_STRICT_SYMCHECK (kb_vanilla_strict.hip): shard invariance against the 4096-filter batch
    of the same rows, bit for bit, and against the oracle on that small batch at 1e-9; a filter made to trip AsSymDense in the
    middle of the batch is the only one flagged and keeps its estimate."""
    N, small, steps = 1 << 20, 4096, 2
    base = synth.linear_batch(small, 6, 3, steps)
    rep = N // small
    big = {kk: (np.tile(v, (rep,) + (1,) * (v.ndim - 1)) if kk != "y" else np.tile(v, (1, rep, 1))) for kk, v in base.items()}
    bad = 117 * small + 33
    p1, p2 = 2.2e20, 5.5e20          # (F P0 F^T)_01 cancels; _01 and _10 keep different multiples of ulp(1e20): tests/test_symcheck_gpu.py
    Fc = np.eye(6); Fc[:2, :2] = [[0.7, -0.7 * 1.3 * p1 / (0.9 * p2)], [1.3, 0.9]]
    Hc = np.zeros((3, 6)); Hc[0, 2] = Hc[1, 3] = Hc[2, 4] = 1.0     # measures the well-scaled states only
    big["F"][bad] = Fc; big["P0"][bad] = np.diag([p1, p2, 1, 1, 1, 1.0]); big["H"][bad] = Hc
    f = orc.Filter.ldkf(orc.VANILLA, big["x0"][bad], big["P0"][bad], big["F"][bad], None, big["H"][bad], big["Q"][bad], big["R"][bad])
    assert f.update(big["y"][0, bad]) == orc.ERR_ASYMMETRIC          # the oracle takes AsSymDense's failure branch for it
    b = ga.FilterBatch.new_ldkf(k.VANILLA, big["x0"], big["P0"], big["F"], None, big["H"], big["Q"], big["R"], flags=k.FLAG_STRICT_SYMCHECK)
    s = ga.FilterBatch.new_ldkf(k.VANILLA, base["x0"], base["P0"], base["F"], None, base["H"], base["Q"], base["R"], flags=k.FLAG_STRICT_SYMCHECK)
    b.update(big["y"][0], snapshot=False); s.update(base["y"][0], snapshot=False)
    st = b.status()
    assert st[bad] & k.ST_ASYMMETRIC and np.count_nonzero(st) == 1
    assert np.array_equal(b.get(k.STATE, bad, 1)[0], big["x0"][bad]) and np.array_equal(b.get(k.COVAR, bad, 1)[0], big["P0"][bad])
    b.update(big["y"][1], snapshot=False); s.update(base["y"][1], snapshot=False)
    tail_P, tail_x = b.get(k.COVAR, N - small, small), b.get(k.STATE, N - small, small)
    assert np.array_equal(tail_P, s.get(k.COVAR)) and np.array_equal(tail_x, s.get(k.STATE))
    xo, Po, _ = orc.ldkf_batch(orc.VANILLA, base["x0"], base["P0"], base["F"], base["H"], base["Q"], base["R"], base["y"][:steps])
    assert synth.rel_frobenius(tail_x, xo) <= 1e-9 and synth.rel_frobenius(tail_P, Po) <= 1e-9


@pytest.mark.parametrize("kind,flags", [(k.VANILLA, 0), (k.SQUAREROOT, 0), (k.INFORMATION, k.FLAG_INFO_FROM_STATE)])
def test_baseline_size_1m_filters_sharing_one_model(kind, flags):
    """1M filters built from ONE model (every model field uploaded with broadcast = 1: the SHARED kernels, model in SGPRs): the last
    4096 filters equal, bit for bit, a 4096-filter batch that was given that model per filter (the oracle-checked path), and the
    first 64 filters match the oracle at 1e-9."""
    N, small, steps = 1 << 20, 4096, 3
    base = synth.linear_batch(small, 6, 3, steps)
    rep = N // small
    F1, H1, Q1, R1 = base["F"][7], base["H"][7], base["Q"][7], base["R"][7]
    tile = lambda M: np.broadcast_to(M, (small,) + M.shape).copy()
    b = ga.FilterBatch.new_ldkf(kind, np.tile(base["x0"], (rep, 1)), np.tile(base["P0"], (rep, 1, 1)), F1, None, H1, Q1, R1, nfilters=N, flags=flags)
    s = ga.FilterBatch.new_ldkf(kind, base["x0"], base["P0"], tile(F1), None, tile(H1), tile(Q1), tile(R1), flags=flags)
    for t in range(steps):
        b.update(np.tile(base["y"][t], (rep, 1)), snapshot=False); s.update(base["y"][t], snapshot=False)
    assert not b.status().any()
    assert np.array_equal(b.get(k.STATE, N - small, small), s.get(k.STATE)) and np.array_equal(b.get(k.COVAR, N - small, small), s.get(k.COVAR))
    assert np.array_equal(b.get(k.COVAR, 0, small), s.get(k.COVAR))
    okind = {k.VANILLA: orc.VANILLA, k.SQUAREROOT: orc.SQUAREROOT, k.INFORMATION: orc.INFORMATION}[kind]
    xo, Po, _ = orc.ldkf_batch(okind, base["x0"][:64], base["P0"][:64], tile(F1)[:64], tile(H1)[:64], tile(Q1)[:64], tile(R1)[:64], base["y"][:steps, :64])
    assert synth.rel_frobenius(b.get(k.STATE, 0, 64), xo) <= 1e-9 and synth.rel_frobenius(b.get(k.COVAR, 0, 64), Po) <= 1e-9
