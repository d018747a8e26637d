"""ctypes binding of the CPU oracle (oracle/gokalman_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  Nothing under gokalman_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GOKALMAN_ORACLE_SO: another build of the same sources (tests/test_sanitizers_cpu.py points it at the ASan / UBSan build)
_SO = os.environ.get("GOKALMAN_ORACLE_SO") or os.path.join(_HERE, "libgokalman_oracle.so")
_SRC = [os.path.join(_HERE, f) for f in ("gokalman_oracle.c", "vanloan_oracle.c", "gokalman_oracle.h")]

VANILLA, VANILLA_PREDICT, SQUAREROOT, INFORMATION, SRIF, HYBRID, BATCH_LS = 1, 2, 3, 4, 5, 6, 7
OK, ERR_SINGULAR, ERR_ASYMMETRIC, ERR_LOCKED, ERR_DIMS, ERR_NOTPD = 0, 1, 2, 3, 4, 5
(GET_STATE, GET_COVAR, GET_PRED_COVAR, GET_GAIN, GET_INNOV, GET_MEAS,
 GET_RAW_VEC, GET_RAW_MAT, GET_RAW_PRED_MAT) = range(9)


def build(force=False):
    """Compile the oracle with gcc if the .so is missing or older than its sources."""
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRC)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_SO)], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.orc_new_ldkf.restype = vp
        L.orc_new_ldkf.argtypes = [C.c_int] * 4 + [_dp] * 7
        L.orc_information_from_state.restype = vp
        L.orc_information_from_state.argtypes = [C.c_int] * 3 + [_dp] * 7
        L.orc_new_srif.restype = vp
        L.orc_new_srif.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int]
        L.orc_new_hybrid.restype = vp
        L.orc_new_hybrid.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_new_batch_ls.restype = vp
        L.orc_new_batch_ls.argtypes = [C.c_int, C.c_int, _dp]
        L.orc_free.argtypes = [vp]
        L.orc_set_state_transition.argtypes = [vp, _dp]
        L.orc_set_input_control.argtypes = [vp, C.c_int, _dp]
        L.orc_set_measurement_matrix.argtypes = [vp, C.c_int, _dp]
        L.orc_set_noise.argtypes = [vp, C.c_int, _dp, _dp]
        L.orc_reset.argtypes = [vp]
        L.orc_update.argtypes = [vp] + [_dp] * 5
        L.orc_prepare.argtypes = [vp, _dp, _dp]
        L.orc_prepare_pnt.argtypes = [vp, _dp]
        L.orc_enable_ekf.argtypes = [vp, C.c_int]
        L.orc_update_nl.argtypes = [vp, _dp, _dp]
        L.orc_predict_nl.argtypes = [vp]
        L.orc_get.argtypes = [vp, C.c_int, _dp]
        L.orc_step.argtypes = [vp]
        L.orc_is_within_nsigma.argtypes = [vp, C.c_double]
        L.orc_inverse.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_cholesky_lower.argtypes = [C.c_int, _dp, _dp]
        L.orc_qr_r.argtypes = [C.c_int, C.c_int, _dp, _dp]
        L.orc_as_sym_dense.argtypes = [C.c_int, _dp, _dp]
        L.orc_sign.restype = C.c_double
        L.orc_sign.argtypes = [C.c_double]
        L.orc_householder_transf.argtypes = [_dp, C.c_int, C.c_int]
        L.orc_measurement_srif_update.argtypes = [C.c_int, C.c_int] + [_dp] * 7
        L.orc_ldkf_batch.restype = C.c_long
        L.orc_ldkf_batch.argtypes = [C.c_int, C.c_long, C.c_int, C.c_int, C.c_int] + [_dp] * 7 + [C.c_int, C.c_int]
        L.orc_vanilla_batch.restype = C.c_long
        L.orc_vanilla_batch.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int] + [_dp] * 7 + [C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_mc_mean_stddev.argtypes = [C.c_long, C.c_int, _dp, _dp, _dp]
        L.orc_smooth_all.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
        L.orc_expm.argtypes = [C.c_int, _dp, _dp]
        L.orc_eigvals.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_van_loan.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp, _dp]
        _lib = L
    return _lib


def _a(x):
    """float64 C-contiguous copy + pointer (None -> NULL)."""
    if x is None:
        return None, None
    arr = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    return arr, arr.ctypes.data_as(_dp)


def _p(x):
    return _a(x)[1] if x is not None else None


class Filter:
    """One reference filter object (any kind), driven in reference call order."""

    def __init__(self, handle, kind, n, p):
        if not handle:
            raise ValueError("oracle constructor failed (dims or not positive definite)")
        self._h = handle
        self.kind, self.n, self.p = kind, n, p

    # --- constructors -----------------------------------------------------
    @classmethod
    def ldkf(cls, kind, x0, P0, F, G, H, Q, R):
        x0, P0, F, H, Q, R = [np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (x0, P0, F, H, Q, R)]
        n = x0.size
        H = H.reshape(-1, n)
        p = H.shape[0]
        m = 0 if G is None else np.asarray(G, dtype=np.float64).reshape(n, -1).shape[1]
        keep = [_a(v) for v in (x0, P0, F, G, H, Q, R)]
        h = lib().orc_new_ldkf(kind, n, p, m, *[k[1] for k in keep])
        return cls(h, kind, n, p)

    @classmethod
    def information_from_state(cls, x0, P0, F, G, H, Q, R):
        x0 = np.asarray(x0, dtype=np.float64)
        n = x0.size
        H = np.asarray(H, dtype=np.float64).reshape(-1, n)
        p = H.shape[0]
        m = 0 if G is None else np.asarray(G, dtype=np.float64).reshape(n, -1).shape[1]
        keep = [_a(v) for v in (x0, P0, F, G, H, Q, R)]
        h = lib().orc_information_from_state(n, p, m, *[k[1] for k in keep])
        return cls(h, INFORMATION, n, p)

    @classmethod
    def srif(cls, x0, P0, R, p, non_tri_r=False):
        n = np.asarray(x0).size
        keep = [_a(v) for v in (x0, P0, R)]
        h = lib().orc_new_srif(n, p, keep[0][1], keep[1][1], keep[2][1], int(non_tri_r))
        return cls(h, SRIF, n, p)

    @classmethod
    def hybrid(cls, x0, P0, Q, R, p):
        n = np.asarray(x0).size
        nq = 0 if Q is None else int(round(np.sqrt(np.asarray(Q).size)))
        keep = [_a(v) for v in (x0, P0, Q, R)]
        h = lib().orc_new_hybrid(n, p, keep[0][1], keep[1][1], nq, keep[2][1], keep[3][1])
        return cls(h, HYBRID, n, p)

    @classmethod
    def batch_ls(cls, n, p, R):
        keep = _a(R)
        return cls(lib().orc_new_batch_ls(n, p, keep[1]), BATCH_LS, n, p)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_free(self._h)
            self._h = None

    # --- LDKF -----------------------------------------------------------------
    def set_state_transition(self, F):
        lib().orc_set_state_transition(self._h, _p(F))

    def set_input_control(self, G):
        G = np.asarray(G, dtype=np.float64).reshape(self.n, -1)
        lib().orc_set_input_control(self._h, G.shape[1], _p(G))

    def set_measurement_matrix(self, H):
        H = np.asarray(H, dtype=np.float64).reshape(-1, self.n)
        self.p = H.shape[0]
        lib().orc_set_measurement_matrix(self._h, self.p, _p(H))

    def set_noise(self, Q, R):
        R = np.atleast_2d(np.asarray(R, dtype=np.float64))
        return lib().orc_set_noise(self._h, R.shape[0], _p(Q), _p(R))

    def reset(self):
        lib().orc_reset(self._h)

    def update(self, y, u=None, w_pred=None, v_meas=None, w_post=None):
        return lib().orc_update(self._h, _p(np.atleast_1d(y)), _p(u), _p(w_pred), _p(v_meas), _p(w_post))

    # --- NLDKF ----------------------------------------------------------------
    def prepare(self, Phi, Htilde):
        lib().orc_prepare(self._h, _p(Phi), _p(Htilde))

    def prepare_pnt(self, Gamma):
        lib().orc_prepare_pnt(self._h, _p(Gamma))

    def enable_ekf(self, on=True):
        lib().orc_enable_ekf(self._h, int(on))

    def update_nl(self, real_obs, computed_obs):
        return lib().orc_update_nl(self._h, _p(real_obs), _p(computed_obs))

    def predict_nl(self):
        return lib().orc_predict_nl(self._h)

    # --- Estimate getters -------------------------------------------------------
    def _get(self, what, shape):
        out = np.zeros(shape, dtype=np.float64)
        rc = lib().orc_get(self._h, what, out.ctypes.data_as(_dp))
        if rc != OK:
            raise RuntimeError("oracle getter failed rc=%d" % rc)
        return out

    def state(self):
        return self._get(GET_STATE, (self.n,))

    def covariance(self):
        return self._get(GET_COVAR, (self.n, self.n))

    def pred_covariance(self):
        return self._get(GET_PRED_COVAR, (self.n, self.n))

    def gain(self):
        return self._get(GET_GAIN, (self.n, self.p))

    def innovation(self):
        size = self.n if self.kind in (INFORMATION, SRIF) else self.p
        return self._get(GET_INNOV, (size,))

    def measurement(self):
        return self._get(GET_MEAS, (self.p,))

    def raw_vec(self):
        return self._get(GET_RAW_VEC, (self.n,))

    def raw_mat(self):
        return self._get(GET_RAW_MAT, (self.n, self.n))

    def raw_pred_mat(self):
        return self._get(GET_RAW_PRED_MAT, (self.n, self.n))

    def step(self):
        return lib().orc_step(self._h)

    def is_within_nsigma(self, N):
        return bool(lib().orc_is_within_nsigma(self._h, float(N)))


# --- primitives -----------------------------------------------------------------
def inverse(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    out = np.zeros_like(A)
    cond = C.c_double(0)
    rc = lib().orc_inverse(n, _p(A), out.ctypes.data_as(_dp), C.byref(cond))
    return rc, out, cond.value


def cholesky_lower(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    out = np.zeros_like(A)
    rc = lib().orc_cholesky_lower(A.shape[0], _p(A), out.ctypes.data_as(_dp))
    return rc, out


def qr_r(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    out = np.zeros_like(A)
    lib().orc_qr_r(A.shape[0], A.shape[1], _p(A), out.ctypes.data_as(_dp))
    return out


def as_sym_dense(M):
    M = np.ascontiguousarray(M, dtype=np.float64)
    out = np.zeros_like(M)
    rc = lib().orc_as_sym_dense(M.shape[0], _p(M), out.ctypes.data_as(_dp))
    return rc, out


def sign(v):
    return lib().orc_sign(float(v))


def householder_transf(A, n, m):
    A = np.ascontiguousarray(A, dtype=np.float64).copy()
    lib().orc_householder_transf(A.ctypes.data_as(_dp), n, m)
    return A


def measurement_srif_update(R, H, b, y):
    R = np.ascontiguousarray(R, dtype=np.float64)
    H = np.ascontiguousarray(H, dtype=np.float64)
    n, m = R.shape[0], H.shape[0]
    Rk, bk, ek = np.zeros((n, n)), np.zeros(n), np.zeros(m)
    lib().orc_measurement_srif_update(n, m, _p(R), _p(H), _p(b), _p(y),
                                      Rk.ctypes.data_as(_dp), bk.ctypes.data_as(_dp), ek.ctypes.data_as(_dp))
    return Rk, bk, ek


def ldkf_batch(kind, x, P, F, H, Q, R, y, threads=None, steps=None):
    """N independent LDKF filters x T steps (AoS inputs); returns (x, P, nerr).
    steps > len(y) cycles through y (step k uses y[k % len(y)])."""
    x = np.array(x, dtype=np.float64, order="C")
    P = np.array(P, dtype=np.float64, order="C")
    N, n = x.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    T, _, p = y.shape
    F, H, Q, R = [np.ascontiguousarray(v, dtype=np.float64) for v in (F, H, Q, R)]
    if threads is None:
        threads = lib().orc_max_threads()
    nerr = lib().orc_ldkf_batch(kind, N, T if steps is None else int(steps), n, p,
                                x.ctypes.data_as(_dp), P.ctypes.data_as(_dp),
                                _p(F), _p(H), _p(Q), _p(R), _p(y), T, int(threads))
    return x, P, nerr


def max_threads():
    return lib().orc_max_threads()


def mc_mean_stddev(states):
    states = np.ascontiguousarray(states, dtype=np.float64)
    runs, n = states.shape
    mean, std = np.zeros(n), np.zeros(n)
    lib().orc_mc_mean_stddev(runs, n, _p(states), mean.ctypes.data_as(_dp), std.ctypes.data_as(_dp))
    return mean, std


def chisquare(kf_factory, truth_states, truth_meas, controls, with_nees=True, with_nis=True):
    """chisquare.go:16-95 NewChiSquare, restated over oracle filters.

    truth_states[run][step][n], truth_meas[run][step][p] are the Monte-Carlo estimates' State() and
    Measurement(); kf_factory() returns a fresh Vanilla oracle filter (the reference Reset()s one filter
    per run, chisquare.go:39).  Returns (NISmeans, NEESmeans) (stat.Mean over runs, :85-94)."""
    truth_states = np.asarray(truth_states, dtype=np.float64)
    truth_meas = np.asarray(truth_meas, dtype=np.float64)
    runs, steps, n = truth_states.shape
    nis = np.zeros((steps, runs))
    nees = np.zeros((steps, runs))
    for r in range(runs):
        kf = kf_factory()
        H = None
        for t in range(steps):
            u = None if controls is None else controls[t if len(controls) > 1 else 0]
            rc = kf.update(truth_meas[r, t], u)
            assert rc == OK, rc
            if with_nees:
                rcode, Pinv, _ = inverse(kf.covariance())
                d = truth_states[r, t] - kf.state()
                nees[t, r] = d @ (Pinv @ d)
            if with_nis:
                # Pyy = H P- H^T + R: the innovation covariance the update inverted for the gain
                Pm, K, innov = kf.pred_covariance(), kf.gain(), kf.innovation()
                H = kf._H
                Pyy = H @ (Pm @ H.T) + kf._R
                rcode, PyyInv, _ = inverse(Pyy)
                nis[t, r] = innov @ (PyyInv @ innov)
    return nis.mean(axis=1), nees.mean(axis=1)


def smooth_all(Phi, x_last, P_last):
    """SmoothAll (hybrid.go:209-238): Phi[steps][n][n] (the STM stored in each estimate), final state / covariance.
    Returns (rc, x[steps][n], P[steps][n][n])."""
    Phi = np.ascontiguousarray(Phi, dtype=np.float64)
    steps, n, _ = Phi.shape
    x = np.zeros((steps, n)); P = np.zeros((steps, n, n))
    x[-1], P[-1] = x_last, P_last
    rc = lib().orc_smooth_all(n, steps, _p(Phi), x.ctypes.data_as(_dp), P.ctypes.data_as(_dp))
    return rc, x, P


def expm(A):
    """mat64.Dense.Exp (Higham 2005 scaling and squaring)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    E = np.zeros_like(A)
    lib().orc_expm(A.shape[0], _p(A), E.ctypes.data_as(_dp))
    return E


def eigvals(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    wr, wi = np.zeros(n), np.zeros(n)
    rc = lib().orc_eigvals(n, _p(A), wr.ctypes.data_as(_dp), wi.ctypes.data_as(_dp))
    return rc, wr + 1j * wi


def van_loan(A, Gamma, W, dt):
    """VanLoan(A, Gamma, W, dt) (c2d.go:13-75) -> (rc, F, Q); rc bit 0 = Nyquist error, bit 1 = Q asymmetric."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    Gamma = np.ascontiguousarray(Gamma, dtype=np.float64).reshape(n, -1)
    q = Gamma.shape[1]
    W = np.ascontiguousarray(W, dtype=np.float64).reshape(q, q)
    F, Q = np.zeros((n, n)), np.zeros((n, n))
    rc = lib().orc_van_loan(n, q, _p(A), _p(Gamma), _p(W), float(dt), F.ctypes.data_as(_dp), Q.ctypes.data_as(_dp))
    return rc, F, Q
