"""AsSymDense's failure branch (helper.go:65-84) on the filter kernels: vanilla.go:207-215 and hybrid.go:183-200 return
(nil, err) when |M_ij - M_ji| exceeds BOTH tolerances of floats.EqualWithinAbsOrRel(.., 1e-6, 1e-2).

In exact arithmetic F P F^T + Q is symmetric; the branch fires on rounding: with P ~ 1e20 and an off-diagonal element of
F P F^T that cancels, M_01 and M_10 are left with DIFFERENT multiples of ulp(1e20) = 16384 because the two entries
multiply the same factors in a different order.  KB_FLAG_STRICT_SYMCHECK runs the statement-by-statement kernel (both
triangles, no FMA contraction), which rounds exactly like gonum / the oracle and therefore takes the branch for exactly the
same filters.  The register kernels compute the upper triangle only (what AsSymDense RETURNS: NewSymDense reads the upper
triangle), so on the same inputs they carry on with the mirrored upper triangle -- documented in DESIGN.md, checked below."""
import numpy as np
import pytest

import gokalman_amd as ga
from gokalman_amd import _capi as k
from gokalman_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _cancelling_models(nfilters, seed):
    """Per filter: F (3x3) whose (0,1) element of F P0 F^T cancels, P0 = diag(p1, p2, 1) with p ~ 1e19..1e21."""
    rng = np.random.default_rng(seed)
    F = np.zeros((nfilters, 3, 3)); P0 = np.zeros((nfilters, 3, 3))
    for i in range(nfilters):
        F00, F10, F11 = rng.uniform(0.5, 1.5, 3)
        p1, p2 = 10 ** rng.uniform(19, 21, 2)
        F[i] = [[F00, -F00 * F10 * p1 / (F11 * p2), 0], [F10, F11, 0], [0, 0, 1.0]]
        P0[i] = np.diag([p1, p2, 1.0])
    return F, P0


def test_vanilla_strict_symcheck_fails_exactly_where_the_oracle_does():
    N = 192
    F, P0 = _cancelling_models(N, seed=5)
    # every third filter is an ordinary, well-scaled one
    d = synth.linear_batch(N, 4, 1, 2)
    F[::3] = np.eye(3) + 0.05 * d["F"][::3, :3, :3]
    P0[::3] = np.diag([2.0, 1.0, 0.5])
    x0 = d["x0"][:, :3].copy()
    H = np.array([[0, 0, 1.0]]); Q = np.diag([1e-3] * 3); R = np.array([[0.1]])
    y = d["y"][:, :, :1]
    want = np.zeros(N, dtype=int); xs = []; Ps = []
    for i in range(N):
        f = orc.Filter.ldkf(orc.VANILLA, x0[i], P0[i], F[i], None, H, Q, R)
        want[i] = f.update(y[0, i])
        xs.append(f.state()); Ps.append(f.covariance())
    nfail = int(np.count_nonzero(want == orc.ERR_ASYMMETRIC))
    assert 30 <= nfail <= 100 and set(want.tolist()) == {orc.OK, orc.ERR_ASYMMETRIC}, (nfail, set(want.tolist()))

    b = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, None, H, Q, R, flags=k.FLAG_STRICT_SYMCHECK)
    b.update(y[0], snapshot=False)
    st = b.status()
    assert np.array_equal(st != 0, want == orc.ERR_ASYMMETRIC)          # the same filters, and only those
    assert set(st[st != 0].tolist()) == {k.ST_ASYMMETRIC}
    bad, good = np.nonzero(st)[0], np.nonzero(st == 0)[0]
    X, P = b.get(k.STATE), b.get(k.COVAR)
    assert np.array_equal(X[bad], x0[bad]) and np.array_equal(P[bad], P0[bad])   # (nil, err): the estimate is untouched
    assert synth.rel_frobenius(X[good], np.array(xs)[good]) <= 1e-9
    assert synth.rel_frobenius(P[good], np.array(Ps)[good]) <= 1e-9

    # the register kernel on the same inputs: upper triangle only, so no asymmetry to see; it proceeds like
    # NewSymDense(n, copy) would (mirror of the upper triangle) and flags nothing finite
    r = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, None, H, Q, R)
    r.update(y[0], snapshot=False)
    assert not r.status().any() and np.isfinite(r.get(k.COVAR)).all()
    assert synth.rel_frobenius(r.get(k.STATE)[good], np.array(xs)[good]) <= 1e-9


def test_hybrid_strict_symcheck_fails_exactly_where_the_oracle_does():
    N = 128
    Phi3, P3 = _cancelling_models(N, seed=9)
    n, p = 6, 2
    Phi = np.tile(np.eye(n), (N, 1, 1)); Phi[:, :3, :3] = Phi3
    P0 = np.tile(np.eye(n), (N, 1, 1)); P0[:, :3, :3] = P3
    Phi[::4] = np.eye(n) + 1e-2 * np.random.default_rng(1).standard_normal((len(Phi[::4]), n, n))
    P0[::4] = np.diag([10.0, 10, 10, 1, 1, 1])
    rng = np.random.default_rng(2)
    x0 = rng.standard_normal((N, n))
    Ht = np.zeros((N, p, n)); Ht[:, 0, 3] = 1.0; Ht[:, 1, 4] = 1.0; Ht[:, :, 2] = 0.3
    R = np.diag([1e-2, 1e-2])
    real = rng.standard_normal((N, p)); comp = real + 1e-2 * rng.standard_normal((N, p))
    want = np.zeros(N, dtype=int); xs = []; Ps = []
    for i in range(N):
        f = orc.Filter.hybrid(x0[i], P0[i], None, R, p)
        f.prepare(Phi[i], Ht[i])
        want[i] = f.update_nl(real[i], comp[i])
        xs.append(f.state()); Ps.append(f.covariance())
    nfail = int(np.count_nonzero(want == orc.ERR_ASYMMETRIC))
    assert nfail >= 20 and set(want.tolist()) == {orc.OK, orc.ERR_ASYMMETRIC}, (nfail, set(want.tolist()))
    b = ga.FilterBatch(k.HYBRID, n, p, 0, N, flags=k.FLAG_STRICT_SYMCHECK)
    b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.init()
    b.prepare(Phi, Ht)
    b.update_nl(real, comp, snapshot=False)
    st = b.status()
    assert np.array_equal(st != 0, want == orc.ERR_ASYMMETRIC)
    assert set(st[st != 0].tolist()) == {k.ST_ASYMMETRIC}
    bad, good = np.nonzero(st)[0], np.nonzero(st == 0)[0]
    X, P = b.get(k.STATE), b.get(k.COVAR)
    assert np.array_equal(X[bad], x0[bad]) and np.array_equal(P[bad], P0[bad])
    assert synth.rel_frobenius(X[good], np.array(xs)[good]) <= 1e-9
    assert synth.rel_frobenius(P[good], np.array(Ps)[good]) <= 1e-9


def test_overflowing_covariance_fails_the_step_on_both_kernels_as_in_the_oracle():
    """P0 ~ 5e308 / 10: F P F^T overflows; the reference's Inverse of S = H P- H^T + R reports the non-finite matrix first
    (vanilla.go:164-167), AsSymDense's NaN != NaN comparison (helper.go:75) never gets to run.  Strict and register
    kernels both leave the estimate alone and flag the filter."""
    N = 70
    d = synth.linear_batch(N, 6, 3, 1)
    P0 = d["P0"].copy()
    huge = [3, 64, 69]
    with np.errstate(over="ignore"):
        P0[huge] *= 5e307
    ok = np.setdiff1d(np.arange(N), huge)
    for flags in (k.FLAG_STRICT_SYMCHECK, 0):
        b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], P0, d["F"], None, d["H"], d["Q"], d["R"], flags=flags)
        b.update(d["y"][0], snapshot=False)
        st = b.status()
        assert sorted(np.nonzero(st)[0].tolist()) == huge
        assert all(int(s) & (k.ST_SINGULAR | k.ST_NONFINITE | k.ST_ASYMMETRIC) for s in st[huge])
        assert np.array_equal(b.get(k.STATE)[huge], d["x0"][huge])      # estimate untouched either way
        assert not st[ok].any()
    for i in huge:
        f = orc.Filter.ldkf(orc.VANILLA, d["x0"][i], P0[i], d["F"][i], None, d["H"][i], d["Q"][i], d["R"][i])
        assert f.update(d["y"][0, i]) == orc.ERR_SINGULAR


@pytest.mark.parametrize("n,p,m", [(3, 1, 0), (4, 2, 0), (4, 2, 1), (6, 3, 0), (5, 3, 2), (6, 4, 2), (2, 1, 0)])
@pytest.mark.parametrize("full", [False, True])
def test_strict_register_kernel_is_bit_identical_to_the_statement_kernel(n, p, m, full):
    """KB_FLAG_STRICT_SYMCHECK batches up to 6 / 4 / 2 run kb_vanilla_strict.hip (registers, compile-time sizes, zero padding);
    KB_FLAG_STATEMENT_KERNELS keeps a batch on vanilla_gen_kernel.  Same operand order, no contraction, same pivots: every bit agrees --
    states, covariances, the FULL outputs and the status words, including filters that fail the symmetry test, singular ones
    and non-finite ones."""
    import torch
    N, steps = 300, 4
    rng = np.random.default_rng(100 * n + 10 * p + m)
    d = synth.linear_batch(N, 8, 4, steps, seed=7)
    F = d["F"][:, :n, :n].copy(); P0 = d["P0"][:, :n, :n].copy(); Q = d["Q"][:, :n, :n].copy()
    H = d["H"][:, :p, :n].copy(); R = d["R"][:, :p, :p].copy(); x0 = d["x0"][:, :n].copy()
    if n >= 3:      # a third of the filters trip AsSymDense by cancellation, as in the test above
        Fc, Pc = _cancelling_models(N, seed=3)
        F[1::3, :3, :3] = Fc[1::3]; F[1::3, 3:, :3] = 0; F[1::3, :3, 3:] = 0
        P0[1::3] = 0; P0[1::3, :3, :3] = Pc[1::3]
        for i in range(3, n):
            P0[1::3, i, i] = 1.0
    H[5] = 0.0; R[5] = 0.0                      # singular innovation covariance
    with np.errstate(over="ignore"):
        P0[8] = np.full((n, n), 1e308)           # overflows in F P F^T
    G = rng.standard_normal((N, n, m)) if m else None
    y = torch.from_numpy(np.ascontiguousarray(d["y"][:, :, :p].transpose(0, 2, 1))).cuda()          # [T][p][N]
    u = torch.from_numpy(np.ascontiguousarray(rng.standard_normal((steps, m, N)))).cuda() if m else None
    flags = k.FLAG_STRICT_SYMCHECK | (k.FLAG_FULL_ESTIMATE if full else 0)
    out = []
    for reg in (True, False):
        b = ga.FilterBatch.new_ldkf(k.VANILLA, x0, P0, F, G, H, Q, R, flags=flags | (0 if reg else k.FLAG_STATEMENT_KERNELS))
        per_step = []
        for t in range(steps):
            up = (u[t].data_ptr(), N) if m else (None, 0)
            if reg or t % 2:
                b.update_dev(y[t].data_ptr(), N, *up)
            else:
                b.update_steps_dev(y[t].data_ptr(), N, 1, *up)   # the fused entry point of a strict batch is the statement kernel too
            b.synchronize()
            fields = [k.STATE, k.COVAR] + ([k.PRED_COVAR, k.GAIN, k.INNOVATION, k.MEASUREMENT] if full else [])
            per_step.append([b.get(f).copy() for f in fields] + [b.status().copy()])
        out.append(per_step)
    nbad = 0
    for t in range(steps):
        st = out[0][t][-1]
        assert np.array_equal(st, out[1][t][-1])
        good = st == 0
        for a_, b_ in zip(out[0][t][:-1], out[1][t][:-1]):
            # failed filters keep their previous estimate (bitwise too); FULL outputs of a failed step are not written
            assert np.array_equal(a_[good].view(np.uint64), b_[good].view(np.uint64))
        assert np.array_equal(out[0][t][0].view(np.uint64), out[1][t][0].view(np.uint64))
        assert np.array_equal(out[0][t][1].view(np.uint64), out[1][t][1].view(np.uint64))
        nbad = max(nbad, int(np.count_nonzero(st)))
    st = out[0][0][-1]
    assert st[5] & k.ST_SINGULAR and st[8] != 0
    if n >= 3:
        assert np.count_nonzero(st & k.ST_ASYMMETRIC) >= 20


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("ekf,full", [(False, False), (True, True), (False, True)])
def test_hybrid_strict_register_kernel_is_bit_identical_to_the_statement_kernel(p, ekf, full):
    """kb_hybrid_strict.hip against hybrid_gen_kernel (KB_FLAG_STATEMENT_KERNELS) over a sequence that takes every branch:
    Update, PreparePNT + Update (SNC, q = 3), Predict(), Update -- CKF and EKF, with filters that trip AsSymDense, one singular
    innovation covariance and one overflow.  Every bit of state, covariance, the FULL members and the status words agrees."""
    N, n, q = 200, 6, 3
    rng = np.random.default_rng(40 + 3 * p + ekf)
    Phi3, P3 = _cancelling_models(N, seed=11)
    Phi = np.tile(np.eye(n), (N, 1, 1)) + 1e-2 * rng.standard_normal((N, n, n))
    P0 = np.tile(np.diag([10.0, 10, 10, 1, 1, 1]), (N, 1, 1))
    Phi[1::3] = np.eye(n); Phi[1::3, :3, :3] = Phi3[1::3]
    P0[1::3] = np.eye(n); P0[1::3, :3, :3] = P3[1::3]
    Ht = np.zeros((N, p, n)); Ht[:, :, 2] = 0.3
    for r in range(p):
        Ht[:, r, 3 + r] = 1.0
    R = np.diag([1e-2] * p)
    Ht[7] = 0.0                                     # with R = 0 below for every filter? no: a zero Htilde row set makes S = R only
    with np.errstate(over="ignore"):
        P0[9] = np.full((n, n), 1e308)              # overflows in Phi P Phi^T
    x0 = rng.standard_normal((N, n))
    Gam = rng.standard_normal((N, n, q)); Qs = np.diag([1e-6, 2e-6, 3e-6])
    obs = [(rng.standard_normal((N, p)), rng.standard_normal((N, p))) for _ in range(3)]
    flags = k.FLAG_STRICT_SYMCHECK | (k.FLAG_FULL_ESTIMATE if full else 0)
    outs = []
    for extra in (0, k.FLAG_STATEMENT_KERNELS):
        b = ga.FilterBatch(k.HYBRID, n, p, q, N, flags=flags | extra)
        b.set(k.X, x0, 1); b.set(k.P, P0, 2); b.set(k.R, R, 2, p_rows=p); b.set(k.Q, Qs, 2); b.init()
        if ekf:
            b.enable_ekf()
        seq = []

        def grab():
            fields = [k.STATE, k.COVAR] + ([k.PRED_COVAR, k.GAIN, k.INNOVATION, k.MEASUREMENT] if full else [])
            seq.append([b.get(f).copy() for f in fields] + [b.status().copy()])
        b.prepare(Phi, Ht); b.update_nl(*obs[0], snapshot=False); grab()
        b.prepare(Phi, Ht); b.prepare_pnt(Gam); b.update_nl(*obs[1], snapshot=False); grab()
        b.prepare(Phi, Ht); b.predict_nl(snapshot=False); grab()
        b.prepare(Phi, Ht); b.update_nl(*obs[2], snapshot=False); grab()
        outs.append(seq)
    for t, (a_, b_) in enumerate(zip(*outs)):
        assert np.array_equal(a_[-1], b_[-1]), t
        good = a_[-1] == 0
        for u_, v_ in zip(a_[:-1], b_[:-1]):
            assert np.array_equal(u_[good].view(np.uint64), v_[good].view(np.uint64)), t
        assert np.array_equal(a_[0].view(np.uint64), b_[0].view(np.uint64)) and np.array_equal(a_[1].view(np.uint64), b_[1].view(np.uint64))
    st = outs[0][0][-1]
    assert np.count_nonzero(st & k.ST_ASYMMETRIC) >= 20 and st[9] != 0
