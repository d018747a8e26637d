"""ctypes declarations for include/gokalman_amd.h (the C ABI of libgokalman_amd.so).

Loading fails loudly when the HIP library has not been built: there is no fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgokalman_amd.so")

# enums (kb_kind, kb_dtype, flags, kb_status, status bits, kb_field, kb_noise_kind)
VANILLA, VANILLA_PREDICT, SQUAREROOT, INFORMATION, SRIF, HYBRID, BATCH_LS = 1, 2, 3, 4, 5, 6, 7
F64, F32 = 0, 1
FLAG_FULL_ESTIMATE, FLAG_STRICT_SYMCHECK, FLAG_INFO_FROM_STATE, FLAG_SRIF_NON_TRI_R, FLAG_STATEMENT_KERNELS = 1, 2, 4, 8, 16
OK, ERR_INVALID, ERR_DIMS, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_LOCKED, ERR_NOT_PD = 0, -1, -2, -3, -4, -5, -6, -7
ST_SINGULAR, ST_ASYMMETRIC, ST_NONFINITE, ST_INFO_NOT_INVERTIBLE, ST_NYQUIST = 1, 2, 4, 8, 16
X, P, F, G, H, Q, R = range(7)
STATE, COVAR, PRED_COVAR, GAIN, INNOVATION, MEASUREMENT, RAW_VEC, RAW_MAT, RAW_PRED_MAT = range(16, 25)
NOISE_NOISELESS, NOISE_AWGN, NOISE_BATCH = 0, 1, 2
MC_KEEP_RUNS = 1
MC_KEEP_MAX_BYTES = 8 << 30

_vp, _dp, _i, _i64, _u64 = C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int64, C.c_uint64

SIGNATURES = {
    # name: (restype, [argtypes])
    "kb_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i64, _i, _i, C.c_uint]),
    "kb_destroy": (None, [_vp]),
    "kb_last_error": (C.c_char_p, []),
    "kb_last_kernel": (C.c_char_p, [C.c_void_p]),
    "kb_version": (C.c_char_p, []),
    "kb_device_count": (_i, []),
    "kb_set": (_i, [_vp, _i, _dp, _i64, _i, _i]),
    "kb_set_dev": (_i, [_vp, _i, _vp, _i64, _i]),
    "kb_init": (_i, [_vp]),
    "kb_reset": (_i, [_vp]),
    "kb_update": (_i, [_vp, _dp, _i, _dp, _i]),
    "kb_update_dev": (_i, [_vp, _vp, _i64, _vp, _i64]),
    "kb_update_steps_dev": (_i, [_vp, _vp, _i64, _vp, _i64, _i]),
    "kb_prepare": (_i, [_vp, _dp, _dp, _i64, _i]),
    "kb_prepare_dev": (_i, [_vp, _vp, _vp, _i64]),
    "kb_prepare_pnt": (_i, [_vp, _dp, _i64, _i]),
    "kb_set_ekf": (_i, [_vp, _i]),
    "kb_ekf_enabled": (_i, [_vp]),
    "kb_update_nl": (_i, [_vp, _dp, _i, _dp, _i]),
    "kb_update_nl_dev": (_i, [_vp, _vp, _vp, _i64]),
    "kb_update_nl_steps_dev": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i]),
    "kb_predict_nl": (_i, [_vp]),
    "kb_smooth_all_dev": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "kb_get": (_i, [_vp, _i, _dp, _i64, _i64]),
    "kb_get_dev": (_i, [_vp, _i, _vp, _i64]),
    "kb_get_estimate": (_i, [_vp, _i64, _i64, _vp]),
    "kb_update_estimate": (_i, [_vp, _dp, _i, _dp, _i, _i64, _i64, _vp]),
    "kb_update_nl_estimate": (_i, [_vp, _dp, _i, _dp, _i, _i64, _i64, _vp]),
    "kb_predict_nl_estimate": (_i, [_vp, _i64, _i64, _vp]),
    "kb_get_status": (_i, [_vp, C.POINTER(C.c_uint32), _i64, _i64]),
    "kb_clear_status": (_i, [_vp]),
    "kb_is_within_nsigma": (_i, [_vp, C.c_double, C.POINTER(C.c_uint8), _i64, _i64]),
    "kb_step": (_i64, [_vp]),
    "kb_filter_step": (_i, [_vp, _i64, C.POINTER(_i64)]),
    "kb_calls": (_i64, [_vp]),
    "kb_replicate": (_i, [_vp, _i64, _i64, C.c_uint, C.POINTER(_vp)]),
    "kb_need_ctrl": (_i, [_vp]),
    "kb_meas_dim": (_i, [_vp]),
    "kb_num_filters": (_i64, [_vp]),
    "kb_stream": (_vp, [_vp]),
    "kb_synchronize": (_i, [_vp]),
    "kb_set_noise_kind": (_i, [_vp, _i, _u64]),
    "kb_set_batch_noise": (_i, [_vp, _dp, _i, _dp, _i]),
    "kb_noise_sample": (_i, [_vp, _i64, _i64, _i64, _i, _dp]),
    "kb_noise_normal": (C.c_double, [_u64, _i64, _i64, _i64, _i, _i]),
    "kb_mc_run": (_i, [_vp, _i, _dp, _i, _i64, _dp]),
    "kb_mc_run_ex": (_i, [_vp, _i, _dp, _i, _i64, _dp, C.c_uint]),
    "kb_mc_get_runs": (_i, [_vp, _i64, _i64, _dp, _dp]),
    "kb_chisquare": (_i, [_vp, _vp, _i, _dp, _i, _i64, _i, _i, _i, _dp]),
    "kb_mc_stats": (_i, [_dp, _i, _i, _i64, _dp, _dp]),
    "kb_sharded_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i64, _i, C.POINTER(_i), _i, C.c_uint]),
    "kb_sharded_destroy": (None, [_vp]),
    "kb_sharded_num_shards": (_i, [_vp]),
    "kb_sharded_shard": (_vp, [_vp, _i]),
    "kb_sharded_first": (_i64, [_vp, _i]),
    "kb_sharded_set": (_i, [_vp, _i, _dp, _i64, _i, _i, _i64]),
    "kb_sharded_set_noise_kind": (_i, [_vp, _i, _u64]),
    "kb_sharded_init": (_i, [_vp]),
    "kb_sharded_reset": (_i, [_vp]),
    "kb_sharded_synchronize": (_i, [_vp]),
    "kb_sharded_update": (_i, [_vp, _dp, _i, _dp, _i]),
    "kb_sharded_update_dev": (_i, [_vp, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_i64)]),
    "kb_sharded_get": (_i, [_vp, _i, _dp, _i64, _i64, _i64]),
    "kb_sharded_get_status": (_i, [_vp, C.POINTER(C.c_uint32), _i64, _i64]),
    "kb_sharded_mc_run": (_i, [_vp, _i, _dp, _i, _dp, C.c_uint]),
    "kb_sharded_chisquare": (_i, [_vp, _vp, _i, _dp, _i, _i, _i, _i, _dp]),
    "kb_sharded_used_rccl": (_i, [_vp]),
    "kb_van_loan": (_i, [_i, _i, _i, _i, _i64, _dp, _dp, _dp, _dp, _i, _dp, _dp, C.POINTER(C.c_uint32)]),
    "kb_van_loan_dev": (_i, [_i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
}



class EstimateView(C.Structure):
    """kb_estimate_view (include/gokalman_amd.h)."""
    _fields_ = [("state", _dp), ("covariance", _dp), ("pred_covariance", _dp), ("gain", _dp), ("innovation", _dp),
                ("measurement", _dp), ("status", C.POINTER(C.c_uint32)), ("clear_status", _i)]


_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built (no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "gokalman_amd: %s is missing -- build the HIP extension first "
                "(python -m gokalman_amd.build or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 and this library NEEDs the same SONAME.  Loaded after
        # torch, the library binds torch's copy (the order every test and bench.py had); loaded BEFORE it -- __graft_entry__.build()
        # followed by smoke() in one process -- the system's copy came first and kb_create then saw no device (round 6).  So: torch first.
        try:
            import torch  # noqa: F401
        except Exception:   # a host without PyTorch (the C++ / Go callers never come through here)
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class KalmanError(RuntimeError):
    """A negative kb_status from the C ABI, with kb_last_error() as the message."""

    def __init__(self, code, message):
        super().__init__("%s (kb_status %d)" % (message, code))
        self.code = code
        self.message = message


def check(rc):
    if rc != OK:
        raise KalmanError(rc, lib().kb_last_error().decode("utf-8", "replace"))
    return rc
