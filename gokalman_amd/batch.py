"""Host-side mirror of gokalman's filter interface over a batch of N filters.

Names and argument meaning follow the reference (kalman.go:35-72): `update(measurement,
control)`, `set_state_transition`, `set_measurement_matrix`, `set_noise`, `reset`, and an
`Estimate` with `state() / measurement() / innovation() / covariance() / pred_covariance()
/ is_within_nsigma()`.  Every call goes straight to the C ABI (include/gokalman_amd.h);
nothing is computed here.  Arrays are numpy, shaped `[N, ...]` per filter or `[...]` for a
model shared by every filter (broadcast).
"""
import ctypes as C

import numpy as np

from . import _capi as k

_dp = C.POINTER(C.c_double)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _ptr(a):
    return a.ctypes.data_as(_dp)


SNAPSHOT_MAX_FILTERS = 4096   # update() returns an owning snapshot up to this batch size, a guarded view above it


class StaleEstimateError(RuntimeError):
    """A non-owning Estimate was read after its batch had moved on to a later step."""


class Estimate:
    """The Estimate interface (kalman.go:64-72) for filters [first, first+count) of the batch at ONE step.

    The reference's Update returns a freshly allocated, immutable estimate (vanilla.go:216-218) that callers keep and
    read later (examples/jerkcar/main.go:71-90, montecarlo.go:108-117).  `snapshot=True` gives exactly that: every member
    is downloaded once (kb_get_estimate: one device synchronisation) and owned by this object.  `snapshot=False` is the
    cheap form for batches too large to copy every step: getters download on demand, and raise StaleEstimateError once
    the batch has advanced (never a silent read of a later step); `freeze()` turns it into an owning snapshot."""

    def __init__(self, batch, snapshot=False, first=0, count=None, clear_status=False, via=None):
        """via(view, first, count): a C-ABI call that runs a step AND fills the view (kb_update_estimate & co.: one synchronisation
        for the step and its estimate); without it an owning estimate is a kb_get_estimate of the batch's current state."""
        self._b = batch
        self._first = first
        self._count = batch.N - first if count is None else count
        self._own = None
        if snapshot or via is not None:
            self._download(clear_status, via)
        self._step = batch.step()
        self._calls = batch.calls()

    # ---- owning form -------------------------------------------------------------------
    def _download(self, clear_status=False, via=None):
        b = self._b
        n, p, cnt = b.n, b.meas_dim(), self._count
        info = b.kind in (k.INFORMATION, k.SRIF)
        full = bool(b.flags & k.FLAG_FULL_ESTIMATE)
        lazy = b.kind in (k.SQUAREROOT, k.INFORMATION, k.SRIF, k.BATCH_LS)
        own = {"state": np.zeros((cnt, n)), "covariance": np.zeros((cnt, n, n)), "status": np.zeros(cnt, dtype=np.uint32)}
        if full:
            own["pred_covariance"] = np.zeros((cnt, n, n))
            if not lazy or b.kind == k.SQUAREROOT:
                own["gain"] = np.zeros((cnt, n, p))
            own["measurement"] = np.zeros((cnt, p))
        if info or full:
            own["innovation"] = np.zeros((cnt, n if info else p))
        v = k.EstimateView()
        for name in ("state", "covariance", "pred_covariance", "gain", "innovation", "measurement"):
            if name in own:
                setattr(v, name, _ptr(own[name]))
        v.status = own["status"].ctypes.data_as(C.POINTER(C.c_uint32))
        v.clear_status = 1 if clear_status else 0
        if via is not None:
            k.check(via(C.byref(v), self._first, cnt))
        else:
            k.check(k.lib().kb_get_estimate(b._h, self._first, cnt, C.byref(v)))
        self._own = own

    def freeze(self):
        """Turn a view into an owning snapshot (must still be the batch's current step)."""
        if self._own is None:
            self._check_live()
            self._download()
        return self

    @property
    def owning(self):
        return self._own is not None

    def _check_live(self):
        if self._b.calls() != self._calls:
            raise StaleEstimateError(
                "this Estimate is a view of step %d but the batch is at step %d: ask update(..., snapshot=True) or call "
                "freeze() before the next Update to keep an estimate (vanilla.go:216-218 semantics)" % (self._step, self._b.step()))

    def _get(self, name, field):
        if self._own is not None:
            if name not in self._own:
                raise k.KalmanError(k.ERR_INVALID, "field %d needs a batch created with KB_FLAG_FULL_ESTIMATE" % field)
            return self._own[name]
        self._check_live()
        return self._b.get(field, self._first, self._count)

    def state(self):
        return self._get("state", k.STATE)

    def measurement(self):
        return self._get("measurement", k.MEASUREMENT)

    def innovation(self):
        return self._get("innovation", k.INNOVATION)

    def covariance(self):
        return self._get("covariance", k.COVAR)

    def pred_covariance(self):
        return self._get("pred_covariance", k.PRED_COVAR)

    def gain(self):
        return self._get("gain", k.GAIN)

    def status(self):
        """Per-filter status bits at this step (0 = the reference's `err == nil`)."""
        if self._own is not None:
            return self._own["status"]
        self._check_live()
        return self._b.status(self._first, self._count)

    def is_within_nsigma(self, nsigma):
        """vanilla.go:231-239: |x_i| <= N sqrt(P_ii) for every component."""
        if self._own is not None:
            x, P = self._own["state"], self._own["covariance"]
            with np.errstate(invalid="ignore"):
                d = nsigma * np.sqrt(np.diagonal(P, axis1=1, axis2=2))
                return ~np.any((x > d) | (x < -d), axis=1)
        self._check_live()
        return self._b.is_within_nsigma(nsigma, self._first, self._count)

    def is_within_2sigma(self):
        return self.is_within_nsigma(2.0)

    def string(self, filt=0):
        """<Kind>Estimate.String() of one filter of this estimate (vanilla.go:276-284 and the other kinds' equivalents,
        see strfmt.py).  Members the batch does not keep (no KB_FLAG_FULL_ESTIMATE) print as Go's nil."""
        from . import strfmt
        name = {k.VANILLA: "vanilla", k.VANILLA_PREDICT: "vanilla", k.SQUAREROOT: "squareroot", k.INFORMATION: "information",
                k.SRIF: "srif", k.HYBRID: "hybrid", k.BATCH_LS: "vanilla"}[self._b.kind]

        def member(fn):
            try:
                return fn()[filt]
            except k.KalmanError:
                return None
        return strfmt.estimate_string(name, member(self.state), member(self.measurement), member(self.covariance),
                                      member(self.gain), member(self.pred_covariance), member(self.innovation))

    def __str__(self):
        return "\n".join(self.string(i) for i in range(self._count))


class FilterBatch:
    """N independent filters of one kind on one MI355X (a `kb_batch`)."""

    def __init__(self, kind, n, p, m=0, nfilters=1, dtype=k.F64, device=0, flags=0):
        self._h = C.c_void_p()
        self.kind, self.n, self.pmax, self.m, self.N, self.dtype = kind, n, p, m, int(nfilters), dtype
        self.flags, self._resets = flags, 0
        self._noise_name = "noiseless"
        k.check(k.lib().kb_create(C.byref(self._h), kind, n, p, m, int(nfilters), dtype, device, flags))

    # ---- constructors mirroring NewVanilla / NewPurePredictorVanilla / NewSquareRoot /
    # ---- NewInformation / NewInformationFromState (x0, P0, F, G, H, noise)
    @classmethod
    def new_ldkf(cls, kind, x0, P0, F, G, H, Q, R, nfilters=None, dtype=k.F64, device=0, flags=0,
                 noise=k.NOISE_NOISELESS, seed=0, pmax=None):
        """pmax: largest measurement dimension a later set_measurement_matrix may use."""
        x0, P0, F, H, Q, R = [_f64(v) for v in (x0, P0, F, H, Q, R)]
        n = x0.shape[-1]
        H = H.reshape(H.shape[:-2] + H.shape[-2:]) if H.ndim >= 2 else H.reshape(1, n)
        p = H.shape[-2]
        if R.ndim < 2:
            R = R.reshape(-1, p, p) if R.size != p * p else R.reshape(p, p)
        Gm = None if G is None else _f64(G)
        m = 0 if Gm is None else (Gm.shape[-1] if Gm.ndim >= 2 else 1)
        if Gm is not None and Gm.ndim == 1:
            Gm = Gm.reshape(n, 1)
        # the reference's constructor checks (vanilla.go:23-31), same messages
        if P0.shape[-1] != n:
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: x0(%dx...) Covar0(...x%d)" % (n, P0.shape[-1]))
        if F.shape[-2] != P0.shape[-1]:
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: F(%dx...) Covar0(...x%d)" % (F.shape[-2], P0.shape[-1]))
        if H.shape[-1] != n:
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: H(...x%d) x0(%dx...)" % (H.shape[-1], n))
        if nfilters is None:
            nfilters = x0.shape[0] if x0.ndim == 2 else 1
        b = cls(kind, n, max(p, pmax or p), m, nfilters, dtype, device, flags)
        b.set(k.X, x0, 1)
        b.set(k.P, P0, 2)
        b.set(k.F, F, 2)
        if Gm is not None and m > 0:
            b.set(k.G, Gm, 2)
        b.set(k.H, H, 2, p_rows=p)
        b.set(k.Q, Q, 2)
        b.set(k.R, R, 2, p_rows=p)
        if noise != k.NOISE_NOISELESS:
            b.set_noise_kind(noise, seed)
        b.init()
        return b

    def replicate(self, nfilters, filt=0, flags=None):
        """kb_replicate: a new batch of `nfilters` copies of filter `filt` (model, INITIAL estimate, noise selection)."""
        flags = (self.flags & (k.FLAG_FULL_ESTIMATE | k.FLAG_STRICT_SYMCHECK)) if flags is None else flags
        h = C.c_void_p()
        k.check(k.lib().kb_replicate(self._h, int(filt), int(nfilters), flags, C.byref(h)))
        b = object.__new__(FilterBatch)
        b._h = h
        b.kind, b.n, b.pmax, b.m, b.N, b.dtype = self.kind, self.n, self.pmax, self.m, int(nfilters), self.dtype
        b.flags = flags | (self.flags & (k.FLAG_INFO_FROM_STATE | k.FLAG_SRIF_NON_TRI_R))
        b._resets, b._noise_name = 0, self._noise_name
        return b

    def __del__(self):
        self.close()

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                k.lib().kb_destroy(h)
            except Exception:  # interpreter shutdown: module globals already torn down
                pass
            self._h = C.c_void_p()

    # ---- uploads -------------------------------------------------------------------
    def _item_shape(self, field, p_rows):
        n, m, p = self.n, self.m, (p_rows or self.pmax)
        q = self.m if self.kind == k.HYBRID else n
        return {k.X: (n,), k.P: (n, n), k.F: (n, n), k.G: (n, m), k.H: (p, n), k.Q: (q, q), k.R: (p, p)}.get(field)

    def set(self, field, arr, item_ndim, p_rows=0):
        """kb_set: arr is [N, ...item] per filter or [...item] shared (broadcast).  The C ABI carries no element
        count, so the item shape is checked here against the batch dimensions."""
        arr = _f64(arr)
        if arr.ndim == item_ndim:
            count, bcast = 1, 1
            item = arr.shape
        elif arr.ndim == item_ndim + 1:
            count, bcast = arr.shape[0], 0
            item = arr.shape[1:]
            if count == 1 and self.N != 1:
                bcast = 1
        else:
            raise ValueError("array rank %d does not match field rank %d" % (arr.ndim, item_ndim))
        want = self._item_shape(field, p_rows)
        if want is not None and tuple(item) != want:
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: field %d expects items of shape %s, got %s" % (field, want, tuple(item)))
        if not bcast and count != self.N:
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: field %d needs 1 or N=%d items, got %d" % (field, self.N, count))
        k.check(k.lib().kb_set(self._h, field, _ptr(arr), count, bcast, p_rows))

    def set_dev(self, field, ptr, ld, p_rows=0):
        k.check(k.lib().kb_set_dev(self._h, field, C.c_void_p(ptr), ld, p_rows))

    def init(self):
        k.check(k.lib().kb_init(self._h))

    # ---- LDKF setters (kalman.go:41-44) ----------------------------------------------
    def set_state_transition(self, F):
        self.set(k.F, F, 2)

    def set_input_control(self, G):
        self.set(k.G, G, 2)

    def set_measurement_matrix(self, H):
        H = _f64(H)
        self.set(k.H, H, 2, p_rows=H.shape[-2])

    def set_noise(self, Q, R):
        """SetNoise(Noiseless/AWGN{Q,R})."""
        R = _f64(R)
        if R.ndim < 2:
            R = R.reshape(1, 1)
        self.set(k.Q, Q, 2)
        self.set(k.R, R, 2, p_rows=R.shape[-1])

    def set_noise_kind(self, kind, seed=0):
        k.check(k.lib().kb_set_noise_kind(self._h, kind, seed))
        self._noise_name = {k.NOISE_NOISELESS: "noiseless", k.NOISE_AWGN: "awgn", k.NOISE_BATCH: "batch"}.get(kind, "noiseless")

    def set_batch_noise(self, process, measurement):
        """SetNoise(BatchNoise{process, measurement}) (noise.go:67-106); Q and R become zero, as BatchNoise reports them."""
        pr, me = _f64(process), _f64(measurement)
        if pr.ndim != 2 or pr.shape[1] != self.n or me.ndim != 2 or me.shape[1] != self.meas_dim():
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: BatchNoise needs process [steps, %d] and measurement [steps, %d]"
                                % (self.n, self.meas_dim()))
        k.check(k.lib().kb_set_batch_noise(self._h, _ptr(pr), pr.shape[0], _ptr(me), me.shape[0]))

    def reset(self):
        k.check(k.lib().kb_reset(self._h))
        self._resets += 1

    # ---- the hot path ----------------------------------------------------------------
    def _per_filter(self, v, rows_name, expect_rows=None):
        """A vector argument as [N, rows] float64: one vector (broadcast to every filter) or exactly N of them.  The C ABI
        reads N * rows doubles from the pointer, so any other leading dimension is rejected here."""
        v = _f64(v)
        if v.ndim == 1:
            v = np.broadcast_to(v, (self.N, v.shape[0]))
        elif v.ndim != 2 or v.shape[0] != self.N:
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: %s must be [rows] or [N=%d, rows], got %s"
                                % (rows_name, self.N, tuple(np.shape(v))))
        return _f64(v)

    def _wants_snapshot(self, snapshot):
        if snapshot is None:   # BatchKF has no per-step estimate: Solve() (kb_get) is an explicit call there (batch.go:64-79)
            snapshot = self.N <= SNAPSHOT_MAX_FILTERS and self.kind != k.BATCH_LS
        return snapshot

    def _estimate(self, snapshot):
        return Estimate(self, snapshot=self._wants_snapshot(snapshot))

    def update(self, measurement, control=None, snapshot=None):
        """LDKF.Update(measurement, control) for every filter; returns the batch Estimate of this step: an owning
        snapshot (the reference's immutable estimate) for batches up to SNAPSHOT_MAX_FILTERS or with snapshot=True,
        a guarded view otherwise (see Estimate)."""
        y = self._per_filter(measurement, "measurement (y)")
        u, urows = None, 0
        if control is not None:
            u = self._per_filter(control, "control (u)")
            urows = u.shape[1]
        up = None if u is None else _ptr(u)
        if self._wants_snapshot(snapshot):   # the step and its estimate in ONE call and one synchronisation
            return Estimate(self, via=lambda view, first, cnt: k.lib().kb_update_estimate(self._h, _ptr(y), y.shape[1], up, urows, first, cnt, view))
        k.check(k.lib().kb_update(self._h, _ptr(y), y.shape[1], up, urows))
        return Estimate(self, snapshot=False)

    def update_dev(self, meas_ptr, ld_meas, ctrl_ptr=None, ld_ctrl=0):
        k.check(k.lib().kb_update_dev(self._h, C.c_void_p(meas_ptr), ld_meas,
                                      C.c_void_p(ctrl_ptr) if ctrl_ptr else None, ld_ctrl))

    def update_steps_dev(self, meas_ptr, ld_meas, nsteps, ctrl_ptr=None, ld_ctrl=0):
        k.check(k.lib().kb_update_steps_dev(self._h, C.c_void_p(meas_ptr), ld_meas,
                                            C.c_void_p(ctrl_ptr) if ctrl_ptr else None, ld_ctrl, nsteps))

    # ---- NLDKF (kalman.go:51-60) -------------------------------------------------------
    def prepare(self, phi, htilde):
        phi, htilde = _f64(phi), _f64(htilde)
        bcast = 1 if phi.ndim == 2 else 0
        lead = () if bcast else (self.N,)
        if phi.shape != lead + (self.n, self.n) or htilde.shape != lead + (self.pmax, self.n):
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: Phi %s Htilde %s for a batch of N=%d, n=%d, p=%d"
                                % (phi.shape, htilde.shape, self.N, self.n, self.pmax))
        k.check(k.lib().kb_prepare(self._h, _ptr(phi), _ptr(htilde), 1 if bcast else phi.shape[0], bcast))

    def prepare_pnt(self, gamma):
        gamma = _f64(gamma)
        bcast = 1 if gamma.ndim == 2 else 0
        if self.kind == k.HYBRID and gamma.shape != (() if bcast else (self.N,)) + (self.n, self.m):
            raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: Gamma %s for a batch of N=%d, n=%d, q=%d"
                                % (gamma.shape, self.N, self.n, self.m))
        k.check(k.lib().kb_prepare_pnt(self._h, _ptr(gamma), 1 if bcast else gamma.shape[0], bcast))

    def enable_ekf(self):
        k.check(k.lib().kb_set_ekf(self._h, 1))

    def disable_ekf(self):
        k.check(k.lib().kb_set_ekf(self._h, 0))

    def ekf_enabled(self):
        return bool(k.lib().kb_ekf_enabled(self._h))

    def update_nl(self, real_obs, computed_obs, snapshot=None):
        r = self._per_filter(real_obs, "real observation")
        c = self._per_filter(computed_obs, "computed observation")
        if self._wants_snapshot(snapshot):
            return Estimate(self, via=lambda view, first, cnt: k.lib().kb_update_nl_estimate(self._h, _ptr(r), r.shape[1], _ptr(c), c.shape[1], first, cnt, view))
        k.check(k.lib().kb_update_nl(self._h, _ptr(r), r.shape[1], _ptr(c), c.shape[1]))
        return Estimate(self, snapshot=False)

    def update_nl_steps_dev(self, phi_ptr, htilde_ptr, ld, phi_step, htilde_step, real_ptr, computed_ptr, ld_obs, obs_step, nsteps):
        """kb_update_nl_steps_dev: nsteps Prepare + Update pairs from one call (planar device arrays, strides in elements)."""
        k.check(k.lib().kb_update_nl_steps_dev(self._h, C.c_void_p(phi_ptr), C.c_void_p(htilde_ptr), ld, phi_step, htilde_step,
                                               C.c_void_p(real_ptr), C.c_void_p(computed_ptr), ld_obs, obs_step, nsteps))

    def predict_nl(self, snapshot=None):
        if self._wants_snapshot(snapshot):
            return Estimate(self, via=lambda view, first, cnt: k.lib().kb_predict_nl_estimate(self._h, first, cnt, view))
        k.check(k.lib().kb_predict_nl(self._h))
        return Estimate(self, snapshot=False)

    # ---- results -----------------------------------------------------------------------
    def _shape(self, field):
        n, p = self.n, self.meas_dim()
        info = self.kind in (k.INFORMATION, k.SRIF)
        return {
            k.STATE: (n,), k.RAW_VEC: (n,), k.X: (n,),
            k.COVAR: (n, n), k.PRED_COVAR: (n, n), k.RAW_MAT: (n, n), k.RAW_PRED_MAT: (n, n), k.P: (n, n),
            k.GAIN: (n, p), k.INNOVATION: (n,) if info else (p,), k.MEASUREMENT: (p,),
            k.F: (n, n), k.H: (p, n), k.G: (n, self.m), k.R: (p, p),
            k.Q: (self.m, self.m) if self.kind == k.HYBRID else (n, n),
        }[field]

    def get(self, field, first=0, count=None):
        count = self.N - first if count is None else count
        out = np.zeros((count,) + self._shape(field), dtype=np.float64)
        k.check(k.lib().kb_get(self._h, field, _ptr(out), first, count))
        return out

    def status(self, first=0, count=None):
        count = self.N - first if count is None else count
        out = np.zeros(count, dtype=np.uint32)
        k.check(k.lib().kb_get_status(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), first, count))
        return out

    def clear_status(self):
        k.check(k.lib().kb_clear_status(self._h))

    def is_within_nsigma(self, nsigma, first=0, count=None):
        count = self.N - first if count is None else count
        out = np.zeros(count, dtype=np.uint8)
        k.check(k.lib().kb_is_within_nsigma(self._h, float(nsigma), out.ctypes.data_as(C.POINTER(C.c_uint8)), first, count))
        return out.astype(bool)

    def estimate(self, snapshot=None):
        """The batch's current estimate (what the reference keeps in kf.prevEst)."""
        return self._estimate(snapshot)

    def step(self):
        """kf.step: not advanced by a failed Update (vanilla.go:164-167).  Exact for filter 0 of a batch of at most 64 filters."""
        return int(k.lib().kb_step(self._h))

    def filter_step(self, filt):
        """kf.step of any filter of the batch (synchronises)."""
        out = C.c_int64()
        k.check(k.lib().kb_filter_step(self._h, int(filt), C.byref(out)))
        return out.value

    def calls(self):
        """Update / Predict / Reset calls accepted so far (monotone)."""
        return int(k.lib().kb_calls(self._h))

    def need_ctrl(self):
        return bool(k.lib().kb_need_ctrl(self._h))

    def meas_dim(self):
        return int(k.lib().kb_meas_dim(self._h))

    def last_kernel(self):
        """The kernel instantiation(s) that served the last step of this batch (kb_last_kernel: a debugging / reporting aid)."""
        return k.lib().kb_last_kernel(self._h).decode("ascii", "replace")

    def stream(self):
        return int(k.lib().kb_stream(self._h) or 0)

    def synchronize(self):
        k.check(k.lib().kb_synchronize(self._h))

    def string(self, filt=0):
        """The filter's String() (vanilla.go:76-78, squareroot.go:65-67, information.go:96-98, hybrid.go:63-65) for filter
        `filt` of the batch, with its noise's String() (noise.go:62-64, :104-106, :162-164); see strfmt.py."""
        from . import strfmt
        name = {k.VANILLA: "vanilla", k.VANILLA_PREDICT: "vanilla", k.SQUAREROOT: "squareroot", k.INFORMATION: "information",
                k.HYBRID: "hybrid"}.get(self.kind)
        if name is None:
            return "gokalman_amd.FilterBatch(kind=%d, n=%d, N=%d)" % (self.kind, self.n, self.N)   # SRIF, BatchKF: no String() in the reference
        noise = strfmt.noise_string(self._noise_name, self.get(k.Q, filt, 1)[0], self.get(k.R, filt, 1)[0])
        if name == "hybrid":
            return strfmt.filter_string(name, None, None, None, noise, self.step())
        F = self.get(k.F, filt, 1)[0]
        if name == "information":
            F = np.linalg.inv(F)   # information.go:96 prints the cached inverse
        G = self.get(k.G, filt, 1)[0] if self.m > 0 else None
        return strfmt.filter_string(name, F, G, self.get(k.H, filt, 1)[0], noise)

    def __str__(self):
        return self.string(0)

    def noise_sample(self, filt, epoch, step, which, size):
        out = np.zeros(size, dtype=np.float64)
        k.check(k.lib().kb_noise_sample(self._h, filt, epoch, step, which, _ptr(out)))
        return out


def _go_f(x):
    """fmt's %f of a float64 (montecarlo.go:71-83, exporter.go): six decimals; Go spells the non-finite values NaN / +Inf / -Inf."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "+Inf" if x > 0 else "-Inf"
    return "%f" % x


class MonteCarloEstimate:
    """MonteCarloRuns.Runs[r].Estimates[k] (montecarlo.go:108-117): the estimate a pure-predictor Vanilla returns
    (vanilla.go:170-179) = {x-, yhat, 0, sym(P-), sym(P-), K}.  State() and Measurement() differ between the runs (kept on
    the device by kb_mc_run_ex(KB_MC_KEEP_RUNS)); the covariances and the gain do not depend on the noise and are shared."""

    def __init__(self, mc, run, step):
        self._mc, self._r, self._k = mc, run, step

    def state(self):
        return self._mc._states()[self._r, self._k]

    def measurement(self):
        return self._mc._measurements()[self._r, self._k]

    def innovation(self):
        return np.zeros(self._mc.p)

    def covariance(self):
        return self._mc._shared()["pred_covariance"][self._k]

    def pred_covariance(self):
        return self._mc._shared()["pred_covariance"][self._k]

    def gain(self):
        return self._mc._shared()["gain"][self._k]

    def is_within_nsigma(self, nsigma):
        x, d = self.state(), nsigma * np.sqrt(np.diagonal(self.covariance()))
        return not bool(np.any((x > d) | (x < -d)))

    def is_within_2sigma(self):
        return self.is_within_nsigma(2.0)


class MonteCarloRun:
    """MonteCarloRun (montecarlo.go:122-124): `Estimates[k]`."""

    class _Estimates:
        def __init__(self, mc, run):
            self._mc, self._r = mc, run

        def __len__(self):
            return self._mc.steps

        def __getitem__(self, k):
            if k < 0:
                k += self._mc.steps
            if not 0 <= k < self._mc.steps:
                raise IndexError(k)
            return MonteCarloEstimate(self._mc, self._r, k)

    def __init__(self, mc, run):
        self.Estimates = MonteCarloRun._Estimates(mc, run)
        self.estimates = self.Estimates


class MonteCarloRuns:
    """MonteCarloRuns (montecarlo.go:11-89): `Runs[r].Estimates[k]`, `Mean(step)`, `StdDev(step)`, `AsCSV(headers)`.

    Mean / StdDev come from per-step sums reduced on the device (kb_mc_run + kb_mc_stats); Runs and AsCSV need the
    trajectories, which the engine keeps only when asked (new_monte_carlo_runs(..., keep_runs=...)) and hands over on first
    use (kb_mc_get_runs)."""

    def __init__(self, runs, steps, n, sums, truth=None, kept=False, first_run=0, template=None, controls=None):
        self.runs, self.steps, self.n = int(runs), int(steps), int(n)
        self.sums = sums  # [steps, 3, n]: sum(x-c), sum((x-c)^2), c
        mean = np.zeros((steps, n))
        std = np.zeros((steps, n))
        k.check(k.lib().kb_mc_stats(_ptr(_f64(sums)), steps, n, runs, _ptr(mean), _ptr(std)))
        self._mean, self._std = mean, std
        self._truth, self._kept, self._first_run, self._template, self._controls = truth, kept, first_run, template, controls
        self.p = truth.meas_dim() if truth is not None else 0
        self._st = self._me = self._sh = None

    def mean(self, step):
        return self._mean[step]

    def stddev(self, step):
        return self._std[step]

    Mean, StdDev = mean, stddev

    # ---- Runs --------------------------------------------------------------------------
    def _need_kept(self):
        if not self._kept or self._truth is None:
            raise k.KalmanError(k.ERR_INVALID, "these Monte-Carlo runs were not kept (new_monte_carlo_runs(..., keep_runs=True)): "
                                "only Mean / StdDev / new_chi_square are available")

    def _download(self):
        self._need_kept()
        N = self._truth.N
        st, me = np.zeros((N, self.steps, self.n)), np.zeros((N, self.steps, self.p))
        k.check(k.lib().kb_mc_get_runs(self._truth._h, 0, N, _ptr(st), _ptr(me)))
        self._st, self._me = st, me

    def _states(self):
        if self._st is None:
            self._download()
        return self._st

    def _measurements(self):
        if self._me is None:
            self._download()
        return self._me

    def _shared(self):
        """P-_k and K_k, k < steps: identical for every run (they do not see the noise) -- one Noiseless filter stepped through
        the controls with KB_FLAG_FULL_ESTIMATE."""
        if self._sh is None:
            self._need_kept()
            one = self._truth.replicate(1, flags=k.FLAG_FULL_ESTIMATE)
            one.set_noise_kind(k.NOISE_NOISELESS)
            P, K = np.zeros((self.steps, self.n, self.n)), np.zeros((self.steps, self.n, self.p))
            y0 = np.zeros(self.p)
            for t in range(self.steps):
                u = None
                if one.need_ctrl():
                    u = np.zeros(one.m) if self._controls is None or len(self._controls) == 1 else self._controls[t]
                est = one.update(y0, u)
                P[t], K[t] = est.pred_covariance()[0], est.gain()[0]
            self._sh = {"pred_covariance": P, "gain": K}
        return self._sh

    @property
    def Runs(self):
        self._need_kept()
        return [MonteCarloRun(self, r) for r in range(self._truth.N)]

    def as_csv(self, headers):
        """MonteCarloRuns.AsCSV(headers) (montecarlo.go:62-89): one string per state component -- a header line
        `h-0,h-1,...,h-mean,h-stddev`, then one line per step with every run's value, the mean and the standard deviation."""
        st = self._states()
        runs = st.shape[0]
        out = []
        for i in range(self.n):
            h = headers[i]
            lines = ["".join("%s-%d," % (h, r) for r in range(runs)) + h + "-mean," + h + "-stddev"]
            for t in range(self.steps):
                lines.append("".join(_go_f(v) + "," for v in st[:, t, i]) + _go_f(self._mean[t, i]) + "," + _go_f(self._std[t, i]))
            out.append("\n".join(lines))
        return out

    AsCSV = as_csv


_AUTO_KEEP_BYTES = 256 << 20   # keep_runs=None keeps ensembles of the reference's size (50 x 120, 15 x 1086 ...), not the benchmark's


def _controls_arg(controls, steps, m):
    """controls []*mat64.Vector (montecarlo.go:98-107): `steps` vectors, or ONE vector standing for zero controls."""
    c = _f64(controls)
    if c.ndim == 1:
        c = c.reshape(1, -1)
    if c.ndim != 2:
        raise k.KalmanError(k.ERR_DIMS, "controls must be [steps][m] or [1][m]")
    if c.shape[0] != 1 and c.shape[0] != steps:
        raise k.KalmanError(k.ERR_INVALID, "must provide as much control vectors as steps, or just one control vector")
    return c


def new_monte_carlo_runs(samples, steps, rows_h, controls, kf, first_run=0, reduce=None, keep_runs=None):
    """NewMonteCarloRuns(samples, steps, rowsH, controls, kf) (montecarlo.go:92-119).

    `kf` is the pure-predictor Vanilla of the reference's call -- ONE filter (FilterBatch with N == 1): the `samples` runs
    the reference performs one after the other on it (Reset() in between) are `samples` copies of it on the device
    (kb_replicate), and kf is left Reset(), as montecarlo.go:116 leaves it.  A FilterBatch that already holds one filter per
    run (N == samples, or a shard of a multi-GPU ensemble together with `first_run` / `reduce`) is used as it is.
    `reduce`, when given, is applied to the per-shard sums (e.g. a torch.distributed all-reduce) and `samples` is then the
    global number of runs.  keep_runs: True keeps every run's State() / Measurement() per step on the device for
    `Runs` / `AsCSV` (refused above KB_MC_KEEP_MAX_BYTES), False keeps only the statistics, None keeps small ensembles."""
    if kf.kind != k.VANILLA_PREDICT:
        raise k.KalmanError(k.ERR_INVALID, "the Kalman filter needed for the Monte Carlo runs must be a pure predictor")
    controls = _controls_arg(controls, steps, kf.m)
    if rows_h != kf.meas_dim():   # montecarlo.go:111 feeds Update a zero vector of rowsH rows: vanilla.go:133-135 rejects any other size
        raise k.KalmanError(k.ERR_DIMS, "dimensions must agree: measurement (y)(%dx...) H(%dx...)" % (rows_h, kf.meas_dim()))
    template = None
    truth = kf
    if kf.N == 1 and samples > 1 and reduce is None:
        template, truth = kf, kf.replicate(samples)
    if keep_runs is None:
        keep_runs = steps * (kf.n + kf.meas_dim()) * truth.N * (8 if kf.dtype == k.F64 else 4) <= _AUTO_KEEP_BYTES
    sums = np.zeros((steps, 3, kf.n), dtype=np.float64)
    k.check(k.lib().kb_mc_run_ex(truth._h, steps, _ptr(controls), controls.shape[0], first_run, _ptr(sums), k.MC_KEEP_RUNS if keep_runs else 0))
    if template is not None:
        template.reset()
    if reduce is not None:  # add the shards' partial sums (rows 0, 1); row 2 (the shift) is identical everywhere
        sums[:, :2, :] = reduce(np.ascontiguousarray(sums[:, :2, :]))
    return MonteCarloRuns(samples, steps, kf.n, sums, truth=truth, kept=bool(keep_runs), first_run=first_run, template=template, controls=controls)


def new_chi_square(kf, runs, controls, with_nees=True, with_nis=True, steps=None, first_run=None, total_runs=None, reduce=None):
    """NewChiSquare(kf, runs, controls, withNEES, withNIS) (chisquare.go:16-95): returns (NISmeans, NEESmeans).

    `runs` is the MonteCarloRuns the truth comes from and `kf` the Vanilla filter under test -- one filter, as in the
    reference, which Reset()s it for every run (chisquare.go:39): the engine replays every run of `runs` against its own copy
    of kf in one launch (kb_chisquare with replay_last_mc: the truth's states and measurements are regenerated from the
    runs' noise streams, so they need not have been kept).  A Vanilla batch with one filter per run is used as it is.
    Beyond the reference: `runs` may be the pure-predictor AWGN FilterBatch itself, for which FRESH runs of `steps` steps
    are drawn.  `reduce` adds the per-shard sums across ranks; `total_runs` is then the global number of runs."""
    if not with_nees and not with_nis:
        raise k.KalmanError(k.ERR_INVALID, "Chi Square requires either NEES or NIS or both")
    if isinstance(runs, MonteCarloRuns):
        truth, replay = runs._truth, 1
        if truth is None:
            raise k.KalmanError(k.ERR_INVALID, "these MonteCarloRuns carry no truth batch (built from sums only)")
        steps = runs.steps
        first_run = runs._first_run if first_run is None else first_run
    elif isinstance(runs, FilterBatch):
        truth, replay = runs, 0
        if steps is None:
            raise ValueError("new_chi_square on a truth batch draws fresh runs: give steps=")
        first_run = 0 if first_run is None else first_run
    else:
        raise TypeError("runs must be the MonteCarloRuns returned by new_monte_carlo_runs (or a pure-predictor AWGN FilterBatch)")
    controls = _f64(controls)
    if controls.ndim == 1:
        controls = controls.reshape(1, -1)
    if controls.shape[0] != 1 and controls.shape[0] != steps:
        raise k.KalmanError(k.ERR_INVALID, "must provide as much control vectors as steps, or just one control vector")
    kfb = kf.replicate(truth.N, flags=0) if (kf.N == 1 and truth.N > 1) else kf
    sums = np.zeros((steps, 2), dtype=np.float64)
    k.check(k.lib().kb_chisquare(truth._h, kfb._h, steps, _ptr(controls), controls.shape[0], first_run,
                                 replay, int(with_nees), int(with_nis), _ptr(sums)))
    if reduce is not None:
        sums = reduce(sums)
    n_runs = float(total_runs if total_runs is not None else truth.N)
    return sums[:, 0] / n_runs, sums[:, 1] / n_runs


def van_loan(A, Gamma, W, dt, dtype=k.F64, device=0):
    """VanLoan(A, Gamma, W, dt) (c2d.go:13-75) for one system or a batch: A [n][n] or [N][n][n], Gamma [..][n][q],
    W [..][q][q], dt scalar or [N].  Returns (F, Q, status) with a leading batch axis iff any input had one;
    status & ST_NYQUIST is the reference's "Nyquist sampling criterion not fulfilled" error value."""
    A, Gamma, W, dt = (np.ascontiguousarray(v, dtype=np.float64) for v in (A, Gamma, W, dt))
    n = A.shape[-1]
    Gamma = Gamma.reshape(Gamma.shape[:-2] + (n, -1)) if Gamma.ndim >= 2 else Gamma.reshape(n, -1)
    q = Gamma.shape[-1]
    batched = [A.ndim == 3, Gamma.ndim == 3, W.ndim == 3, dt.ndim == 1]
    sizes = {v.shape[0] for v, bt in zip((A, Gamma, W, dt), batched) if bt}
    if len(sizes) > 1:
        raise ValueError("van_loan: batched arguments disagree on N: %s" % sorted(sizes))
    N = sizes.pop() if sizes else 1
    if A.shape[-2:] != (n, n) or W.shape[-2:] != (q, q):
        raise ValueError("van_loan: A must be n x n, Gamma n x q, W q x q")
    bc = sum(bit for bit, bt in zip((1, 2, 4, 8), batched) if not bt)
    F, Q = np.zeros((N, n, n)), np.zeros((N, n, n))
    st = np.zeros(N, dtype=np.uint32)
    k.check(k.lib().kb_van_loan(device, dtype, n, q, N, _ptr(A), _ptr(Gamma), _ptr(W), _ptr(dt.reshape(-1)), bc,
                                _ptr(F), _ptr(Q), st.ctypes.data_as(C.POINTER(C.c_uint32))))
    if not any(batched):
        return F[0], Q[0], int(st[0])
    return F, Q, st


class ShardedBatch:
    """kb_sharded_* (include/gokalman_amd.h): N filters in contiguous shards over several devices from ONE process -- one handle,
    host thread and stream per device, no collective in Update, the Monte-Carlo / chi-square statistics one ncclAllReduce (host
    sum when shards share a device).  The native counterpart of one-process-per-GPU over torch.distributed (dist.py)."""

    def __init__(self, kind, x0, P0, F, G, H, Q, R, nfilters, devices=None, flags=0, noise=k.NOISE_NOISELESS, seed=0):
        x0, P0, F, H, Q, R = [_f64(v) for v in (x0, P0, F, H, Q, R)]
        self.n, self.p = x0.shape[-1], H.shape[-2]
        Gm = None if G is None else _f64(G).reshape(_f64(G).shape[:-2] + (self.n, -1))
        self.m = 0 if Gm is None else Gm.shape[-1]
        self.N, self.kind = int(nfilters), kind
        if devices is None:
            devices = list(range(k.lib().kb_device_count()))
        devs = (C.c_int * len(devices))(*devices)
        self._s = C.c_void_p()
        k.check(k.lib().kb_sharded_create(C.byref(self._s), kind, self.n, self.p, self.m, self.N, k.F64, devs, len(devices), flags))
        for field, arr, item_ndim, p_rows in ((k.X, x0, 1, 0), (k.P, P0, 2, 0), (k.F, F, 2, 0), (k.G, Gm, 2, 0), (k.H, H, 2, self.p),
                                              (k.Q, Q, 2, 0), (k.R, R, 2, self.p)):
            if arr is None:
                continue
            shared = arr.ndim == item_ndim
            per = int(np.prod(arr.shape[-item_ndim:]))
            k.check(k.lib().kb_sharded_set(self._s, field, _ptr(arr), 1 if shared else self.N, 1 if shared else 0, p_rows, per))
        if noise != k.NOISE_NOISELESS:
            k.check(k.lib().kb_sharded_set_noise_kind(self._s, noise, seed))
        k.check(k.lib().kb_sharded_init(self._s))

    def close(self):
        s = getattr(self, "_s", None)
        if s is not None and s.value:
            k.lib().kb_sharded_destroy(s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shards(self):
        return int(k.lib().kb_sharded_num_shards(self._s))

    def first(self, g):
        return int(k.lib().kb_sharded_first(self._s, g))

    def update(self, measurement, control=None):
        y = _f64(measurement)
        if y.ndim == 1:
            y = _f64(np.broadcast_to(y, (self.N, y.shape[0])))
        u = None
        if control is not None:
            u = _f64(control)
            if u.ndim == 1:
                u = _f64(np.broadcast_to(u, (self.N, u.shape[0])))
        k.check(k.lib().kb_sharded_update(self._s, _ptr(y), y.shape[1], None if u is None else _ptr(u), 0 if u is None else u.shape[1]))

    def get(self, field, shape):
        out = np.zeros((self.N,) + tuple(shape))
        k.check(k.lib().kb_sharded_get(self._s, field, _ptr(out), 0, self.N, int(np.prod(shape))))
        return out

    def status(self):
        out = np.zeros(self.N, dtype=np.uint32)
        k.check(k.lib().kb_sharded_get_status(self._s, out.ctypes.data_as(C.POINTER(C.c_uint32)), 0, self.N))
        return out

    def monte_carlo(self, steps, controls):
        """NewMonteCarloRuns over every shard: MonteCarloRuns (mean / stddev per step over ALL N runs)."""
        controls = _controls_arg(controls, steps, self.m)
        sums = np.zeros((steps, 3, self.n))
        k.check(k.lib().kb_sharded_mc_run(self._s, steps, _ptr(controls), controls.shape[0], _ptr(sums), 0))
        return MonteCarloRuns(self.N, steps, self.n, sums)

    def chi_square(self, kf, steps, controls, replay_last_mc=True, with_nees=True, with_nis=True):
        controls = _controls_arg(controls, steps, self.m)
        sums = np.zeros((steps, 2))
        k.check(k.lib().kb_sharded_chisquare(self._s, kf._s, steps, _ptr(controls), controls.shape[0], int(replay_last_mc),
                                             int(with_nees), int(with_nis), _ptr(sums)))
        return sums[:, 0] / self.N, sums[:, 1] / self.N

    def used_rccl(self):
        return bool(k.lib().kb_sharded_used_rccl(self._s))
