"""ctypes binding of libqpnet_hip.so (include/qpnet_hip.h).  No torch types cross this boundary:
only raw device pointers (tensor.data_ptr()), sizes and the HIP stream handle."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QPN_LIB") or os.path.join(_HERE, "libqpnet_hip.so")     # QPN_LIB: dev aid (build variants)
_lib = None


class QpnConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_quantize", "n_aux", "n_resch", "n_skipch", "dilationF_depth", "dilationF_repeat",
        "dilationA_depth", "dilationA_repeat", "kernel_size", "upsampling_factor")]


class QpnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libqpnet_hip error %d: %s" % (code, msg))
        self.code = code


# every symbol include/qpnet_hip.h declares: (name, restype, argtypes)
_vp, _i, _i64, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
_DECODE_ARGS = [_vp, _i, _i, _i64, _i64, _vp, _vp, _vp, _i, C.POINTER(C.c_int64), _i, _i, _u64, _vp, _vp, _vp, _vp]
SYMBOLS = [
    ("qpn_version", _i, []),
    ("qpn_last_error", C.c_char_p, []),
    ("qpn_param_count", _i64, [C.POINTER(QpnConfig)]),
    ("qpn_create", _i, [C.POINTER(QpnConfig), C.POINTER(_vp)]),
    ("qpn_destroy", None, [_vp]),
    ("qpn_set_weights", _i, [_vp, _vp, C.c_size_t, _vp]),
    ("qpn_decode", _i, _DECODE_ARGS),
    ("qpn_decode_enqueue", _i, _DECODE_ARGS),
    ("qpn_decode_finish", _i, [_vp, _vp]),
    ("qpn_last_decode_kernel_ms", C.c_float, [_vp]),
    ("qpn_last_decode_plan", C.c_char_p, [_vp]),
    ("qpn_train_forward", _i, [_vp, _vp, _i, _i64, _i64, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("qpn_train_backward", _i, [_vp, _vp, _vp, _vp]),
    ("qpn_train_backward_ex", _i, [_vp, _vp, _vp, C.c_float, _i, _vp]),
    ("qpn_train_generation", _i64, [_vp]),
    ("qpn_train_status", _i, [_vp, _vp]),
    ("qpn_train_status_enqueue", _i, [_vp, _vp]),
    ("qpn_train_status_collect", _i, [_vp]),
    ("qpn_train_status_collect_lagged", _i, [_vp]),
    ("qpn_train_status_poll", _i, [_vp, C.POINTER(C.c_int)]),
    ("qpn_train_forward_loss", _i, [_vp, _vp, _i, _i64, _i64, _i64, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _i, _vp, _vp]),
    ("qpn_train_loss", _i, [_vp, C.POINTER(C.c_double), _vp]),
    ("qpn_train_loss_enqueue", _i, [_vp, _vp]),
    ("qpn_train_loss_collect", _i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("qpn_ce_loss", _i, [_vp, _vp, _vp, _i64, _i, _i, _vp, C.POINTER(C.c_double), _vp]),
    ("qpn_adam_step", _i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    ("qpn_adam_step_ex", _i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp]),
    ("qpn_train_step", _i, [_vp, _vp, _i, _i64, _i64, _i64, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64,
                            _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _i, C.POINTER(C.c_double), C.POINTER(C.c_int), _vp]),
    ("qpn_train_applied_updates", _i, [_vp, C.POINTER(C.c_int64), _vp]),
    ("qpn_train_stack_stats", _i, [_vp, C.POINTER(C.c_uint), _i, _vp]),
    ("qpn_train_early_bucket", _i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _vp]),
    ("qpn_train_early_first", _i64, [_vp]),
    ("qpn_train_profile_begin", _i, [_vp, _vp]),
    ("qpn_train_profile_begin_overlapped", _i, [_vp, _vp]),
    ("qpn_train_profile_mark", _i, [_vp, _i, _vp]),
    ("qpn_train_profile_end", _i, [_vp, C.POINTER(C.c_float), _i, _vp]),
    ("qpn_dilated_index_train", _i, [_vp, _i, _i64, _i, _vp, _vp]),
    ("qpn_dilated_index_gen_f32", _i, [_vp, _i64, _i, _vp, _vp]),
    ("qpn_dilated_index_gen_f64", _i, [_vp, _i64, _i, _vp, _vp]),
]


def lib():
    """Load the HIP library; fails loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). qpnet_amd has no CPU/PyTorch fallback." % LIB_PATH)
        # torch ships its own libamdhip64; load it FIRST so this library binds to the same HIP
        # runtime instance (two runtimes in one process do not see each other's device state).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise QpnError(rc, lib().qpn_last_error().decode("utf-8", "replace"))


def make_config(cfg):
    return QpnConfig(*cfg.as_tuple())
