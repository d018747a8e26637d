"""Seeded random call sequences through the LDKF surface (update / SetMeasurementMatrix with a changing row count /
SetNoise / SetStateTransition / Reset), mirrored call by call on the oracle: the setters' side effects
(squareroot.go:100-114 re-Cholesky, information.go:117-138 F^-1 refresh and stale R^-1) and the register / padded /
generic kernel switches they trigger must never diverge from the reference order of operations."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
from oracle import oracle as orc
from tests.achieved import within

pytestmark = pytest.mark.gpu


def _spd(rng, N, d, scale):
    A = rng.standard_normal((N, d, d))
    return scale * (A @ np.swapaxes(A, 1, 2)) + scale * np.eye(d)


@pytest.mark.parametrize("kind,okind,tol", [(k.VANILLA, orc.VANILLA, 1e-9), (k.SQUAREROOT, orc.SQUAREROOT, 1e-9), (k.INFORMATION, orc.INFORMATION, 1e-8)])
@pytest.mark.parametrize("n,m,seed", [(4, 1, 1), (6, 0, 2), (5, 2, 3), (3, 0, 4)])
def test_random_call_sequences_match_the_oracle(kind, okind, tol, n, m, seed):
    rng = np.random.default_rng(seed)
    N, nops = 70, 40
    F = np.eye(n) + 0.05 * rng.standard_normal((N, n, n))
    G = rng.standard_normal((N, n, m)) if m else None
    Hs = {1: rng.standard_normal((N, 1, n)), 2: rng.standard_normal((N, 2, n))}
    Q = _spd(rng, N, n, 1e-3)
    Rs = {1: _spd(rng, N, 1, 1e-2), 2: _spd(rng, N, 2, 1e-2)}
    x0 = rng.standard_normal((N, n)); P0 = _spd(rng, N, n, 1.0)
    # Information caches R^-1 at construction and never refreshes it: only a 1x1 R^-1 (scalar branch, information.go:198-200)
    # survives a change of H's row count, so that kind starts with the 1-row H like examples/jerkcar does
    p = 1 if kind == k.INFORMATION else 2
    flags = k.FLAG_INFO_FROM_STATE if kind == k.INFORMATION else 0
    b = ga.FilterBatch.new_ldkf(kind, x0, P0, F, G, Hs[p], Q, Rs[p], flags=flags, pmax=2)
    mk = (lambda i: orc.Filter.information_from_state(x0[i], P0[i], F[i], G[i] if m else None, Hs[1][i], Q[i], Rs[1][i])) \
        if kind == k.INFORMATION else (lambda i: orc.Filter.ldkf(okind, x0[i], P0[i], F[i], G[i] if m else None, Hs[2][i], Q[i], Rs[2][i]))
    fs = [mk(i) for i in range(N)]
    nupd = 0
    for _ in range(nops):
        op = rng.choice(["update", "update", "update", "swap_h", "set_noise", "set_f", "reset"])
        if op == "update":
            y = rng.standard_normal((N, p)); u = rng.standard_normal((N, m)) if m else None
            b.update(y, u)
            for i, f in enumerate(fs):
                assert f.update(y[i], u[i] if m else None) == orc.OK
            nupd += 1
        elif op == "swap_h":
            p = 3 - p
            b.set_measurement_matrix(Hs[p]); b.set_noise(Q, Rs[p])      # jerkcar's protocol: H and the noise change together
            for i, f in enumerate(fs):
                f.set_measurement_matrix(Hs[p][i]); f.set_noise(Q[i], Rs[p][i])
        elif op == "set_noise":
            Q = _spd(rng, N, n, 1e-3); Rs[p] = _spd(rng, N, p, 1e-2)
            b.set_noise(Q, Rs[p])
            for i, f in enumerate(fs):
                f.set_noise(Q[i], Rs[p][i])
        elif op == "set_f":
            F = np.eye(n) + 0.05 * rng.standard_normal((N, n, n))
            b.set_state_transition(F)
            for i, f in enumerate(fs):
                f.set_state_transition(F[i])
        else:
            b.reset()
            for f in fs:
                f.reset()
    y = rng.standard_normal((N, p)); u = rng.standard_normal((N, m)) if m else None
    est = b.update(y, u)
    xs, Ps = [], []
    for i, f in enumerate(fs):
        assert f.update(y[i], u[i] if m else None) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance())
    assert within(synth.rel_frobenius(est.state(), np.array(xs)), tol, "state"), nupd
    assert within(synth.rel_frobenius(est.covariance(), np.array(Ps)), tol, "covariance"), nupd
    assert not (b.status() & ~np.uint32(k.ST_INFO_NOT_INVERTIBLE)).any()


def test_information_stale_rinv_shape_mismatch_is_an_error():
    """Information built with a 2x2 R and then given a 1-row H: H^T R^-1 is a shape panic in the reference
    (information.go:197-203, R^-1 is never refreshed); the oracle reports ERR_DIMS, the engine KB_ERR_DIMS."""
    rng = np.random.default_rng(5)
    n = 4
    F = np.eye(n); H2 = rng.standard_normal((2, n)); H1 = rng.standard_normal((1, n))
    b = ga.FilterBatch.new_ldkf(k.INFORMATION, np.zeros(n), np.eye(n), F, None, H2, 1e-3 * np.eye(n), 1e-2 * np.eye(2),
                                flags=k.FLAG_INFO_FROM_STATE, pmax=2)
    f = orc.Filter.information_from_state(np.zeros(n), np.eye(n), F, None, H2, 1e-3 * np.eye(n), 1e-2 * np.eye(2))
    b.set_measurement_matrix(H1); b.set_noise(1e-3 * np.eye(n), np.array([[1e-2]]))
    f.set_measurement_matrix(H1); f.set_noise(1e-3 * np.eye(n), np.array([[1e-2]]))
    assert f.update(np.array([0.5])) == orc.ERR_DIMS
    with pytest.raises(ga.KalmanError, match="dimension mismatch"):
        b.update(np.array([0.5]))


@pytest.mark.parametrize("kind,n,p,dtype,tol", [(k.HYBRID, 6, 2, k.F64, 1e-8), (k.HYBRID, 6, 1, k.F64, 1e-8), (k.HYBRID, 6, 3, k.F64, 1e-8), (k.SRIF, 6, 2, k.F64, 1e-8), (k.SRIF, 12, 6, k.F64, 1e-8),
                                                (k.SRIF, 12, 6, k.F32, 5e-3)])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_nldkf_call_sequences_match_the_oracle(kind, n, p, dtype, tol, seed):
    """Prepare / PreparePNT / Update / Predict / EnableEKF / DisableEKF / Reset in random order (hybrid.go, srif.go): covers the
    host-side bookkeeping behind the kernel choice (SRIF's triangular-R tracking, the SNC / Predict instantiation of the
    Hybrid kernel, lock / unlock)."""
    rng = np.random.default_rng(100 * seed + n)
    N, nops, q = 70, 30, 3
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = np.concatenate([np.full(n // 2, 10.0), np.full(n - n // 2, 1.0)])
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-3), np.log(1e-1), size=(N, p)))
    Aq = rng.standard_normal((N, q, q)); Q = 1e-6 * (Aq @ np.swapaxes(Aq, 1, 2) + np.eye(q))
    hyb = kind == k.HYBRID
    b = ga.FilterBatch(kind, n, p, q if hyb else 0, N, dtype=dtype)
    b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p)
    if hyb:
        b.set(k.Q, Q, 2)
    b.init()
    fs = [orc.Filter.hybrid(x0[i], P0[i], Q[i], R[i], p) if hyb else orc.Filter.srif(x0[i], P0[i], R[i], p) for i in range(N)]
    for _ in range(nops):
        op = rng.choice(["update", "update", "update", "predict", "ekf", "reset", "pnt"])
        if op in ("update", "predict", "pnt"):
            Phi = np.eye(n) + 1e-2 * rng.standard_normal((N, n, n)); Ht = rng.standard_normal((N, p, n))
            b.prepare(Phi, Ht)
            for i, f in enumerate(fs):
                f.prepare(Phi[i], Ht[i])
            if op == "pnt" and hyb:
                Gam = rng.standard_normal((N, n, q))
                b.prepare_pnt(Gam)
                for i, f in enumerate(fs):
                    f.prepare_pnt(Gam[i])
            if op == "predict":
                b.predict_nl()
                for f in fs:
                    assert f.predict_nl() == orc.OK
            else:
                real = rng.standard_normal((N, p)); comp = real + 1e-2 * rng.standard_normal((N, p))
                b.update_nl(real, comp)
                for i, f in enumerate(fs):
                    assert f.update_nl(real[i], comp[i]) == orc.OK
        elif op == "ekf" and hyb:
            on = bool(rng.integers(2))
            b.enable_ekf() if on else b.disable_ekf()
            for f in fs:
                f.enable_ekf(on)
        elif op == "reset":
            b.reset()
            for f in fs:
                f.reset()
    Phi = np.eye(n) + 1e-2 * rng.standard_normal((N, n, n)); Ht = rng.standard_normal((N, p, n))
    real = rng.standard_normal((N, p)); comp = real + 1e-2 * rng.standard_normal((N, p))
    b.prepare(Phi, Ht)
    est = b.update_nl(real, comp)
    xs, Ps = [], []
    for i, f in enumerate(fs):
        f.prepare(Phi[i], Ht[i])
        assert f.update_nl(real[i], comp[i]) == orc.OK
        xs.append(f.state()); Ps.append(f.covariance())
    assert within(synth.rel_frobenius(est.state(), np.array(xs)), tol, "state")
    assert within(synth.rel_frobenius(est.covariance(), np.array(Ps)), tol, "covariance")
    assert not b.status().any()


@pytest.mark.parametrize("kind,n,p,dtype,seed", [(k.SRIF, 12, 6, k.F32, 1), (k.SRIF, 12, 6, k.F32, 2), (k.HYBRID, 6, 2, k.F64, 3), (k.HYBRID, 6, 3, k.F64, 4),
                                                 (k.SRIF, 8, 4, k.F64, 5)])
def test_random_sequences_of_multi_step_single_step_and_predict_calls(kind, n, p, dtype, seed):
    """kb_update_nl_steps_dev (round 6) inside random call sequences -- multi-step calls of random length, single Prepare + Update pairs,
    Predict() (after which the SRIF batch is NOT in its steady state: the next multi-step call must fall back to single launches for its
    first step... or all of them), an occasional singular Phi / non-finite observation -- against a twin batch driven by single calls only:
    the same bits, status words, per-filter kf.step and call count after every segment.  (8 / 4 fp64 has no fused kernel: the call is T launches.)"""
    import torch
    rng = np.random.default_rng(1000 + seed)
    N, ld = 150, 160
    tdt, npdt = (torch.float32, np.float32) if dtype == k.F32 else (torch.float64, np.float64)
    x0 = rng.standard_normal((N, n))
    P0 = np.zeros((N, n, n)); P0[:, np.arange(n), np.arange(n)] = [10.0] * (n // 2) + [1.0] * (n - n // 2)
    R = np.zeros((N, p, p)); R[:, np.arange(p), np.arange(p)] = np.exp(rng.uniform(np.log(1e-4), np.log(1e-2), size=(N, p)))

    def new():
        b = ga.FilterBatch(kind, n, p, 0, N, dtype=dtype)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R if kind == k.SRIF else R[0], 2, p_rows=p); b.init()
        if kind == k.HYBRID and seed % 2:
            b.enable_ekf()
        return b
    a, s = new(), new()

    def planar(arr, T):
        out = torch.full((T, int(np.prod(arr.shape[2:])), ld), float("nan"), dtype=tdt)
        out[:, :, :N] = torch.from_numpy(arr.reshape(T, N, -1).transpose(0, 2, 1).astype(npdt))
        return out.cuda()
    for seg in range(12):
        what = rng.choice(["multi", "single", "predict"], p=[0.5, 0.3, 0.2])
        T = int(rng.integers(2, 6)) if what == "multi" else 1
        Phi = np.eye(n) + 1e-2 * rng.standard_normal((T, N, n, n))
        Ht = rng.standard_normal((T, N, p, n)); real = rng.standard_normal((T, N, p)); comp = real + 1e-2 * rng.standard_normal((T, N, p))
        if rng.random() < 0.4:   # somebody fails somewhere in this segment
            t_bad, f_bad = int(rng.integers(0, T)), int(rng.integers(0, N))
            if kind == k.SRIF:
                Phi[t_bad, f_bad, 1, :] = 0.0
            else:
                real[t_bad, f_bad, 0] = np.nan
        dPhi, dH, dre, dco = planar(Phi, T), planar(Ht, T), planar(real, T), planar(comp, T)
        torch.cuda.synchronize()
        if what == "predict":
            for b in (a, s):
                k.check(k.lib().kb_prepare_dev(b._h, dPhi[0].data_ptr(), dH[0].data_ptr(), ld)); b.predict_nl(snapshot=False)
        else:
            if what == "multi":
                a.update_nl_steps_dev(dPhi.data_ptr(), dH.data_ptr(), ld, n * n * ld, p * n * ld, dre.data_ptr(), dco.data_ptr(), ld, p * ld, T)
            else:
                k.check(k.lib().kb_prepare_dev(a._h, dPhi[0].data_ptr(), dH[0].data_ptr(), ld))
                k.check(k.lib().kb_update_nl_dev(a._h, dre[0].data_ptr(), dco[0].data_ptr(), ld))
            for t in range(T):
                k.check(k.lib().kb_prepare_dev(s._h, dPhi[t].data_ptr(), dH[t].data_ptr(), ld))
                k.check(k.lib().kb_update_nl_dev(s._h, dre[t].data_ptr(), dco[t].data_ptr(), ld))
        a.synchronize(); s.synchronize()
        for field in (k.RAW_MAT, k.RAW_VEC) if kind == k.SRIF else (k.STATE, k.COVAR):
            ga_, gs_ = a.get(field), s.get(field)
            assert np.array_equal(np.isnan(ga_), np.isnan(gs_)) and np.array_equal(np.nan_to_num(ga_).view(np.uint64), np.nan_to_num(gs_).view(np.uint64)), (seg, what, field)
        assert np.array_equal(a.status(), s.status()), (seg, what)
        assert a.calls() == s.calls() and [a.filter_step(i) for i in (0, 7, N - 1)] == [s.filter_step(i) for i in (0, 7, N - 1)], (seg, what)
