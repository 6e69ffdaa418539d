"""ctypes front-end of the CPU oracle (oracle/qpnet_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (qpnet_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_quantize", "n_aux", "n_resch", "n_skipch", "dilF_depth", "dilF_repeat",
        "dilA_depth", "dilA_repeat", "kernel_size", "upsampling_factor")]


def build(force=False):
    """gcc-compile the oracle (both variants). Building the checker is not using it."""
    out = os.path.join(_HERE, "_build", "liboracle_generic.so")
    src = os.path.join(_HERE, "qpnet_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)


def _has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            flags = f.read()
        return " fma " in flags and " avx2 " in flags
    except OSError:
        return False


def lib():
    global _LIB
    if _LIB is None:
        build()
        name = "liboracle_fma.so" if _has_fma() else "liboracle_generic.so"
        L = C.CDLL(os.path.join(_HERE, "_build", name))
        L.qpo_param_count.restype = C.c_int64
        L.qpo_qexp.restype = C.c_float
        L.qpo_qexp.argtypes = [C.c_float]
        L.qpo_qgate.restype = C.c_float
        L.qpo_qgate.argtypes = [C.c_float, C.c_float]
        L.qpo_decode.restype = C.c_int
        _LIB = L
    return _LIB


def _cfg(cfg):
    return _Cfg(*cfg.as_tuple())


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def param_count(cfg):
    return int(lib().qpo_param_count(C.byref(_cfg(cfg))))


def encode_mu_law(x, mu=256):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(x.shape, dtype=np.int64)
    lib().qpo_encode_mu_law(_p(x, C.c_double), C.c_int64(x.size), C.c_int(mu), _p(out, C.c_int64))
    return out


def decode_mu_law(y, mu=256):
    y = np.ascontiguousarray(y, dtype=np.int64)
    out = np.empty(y.shape, dtype=np.float64)
    lib().qpo_decode_mu_law(_p(y, C.c_int64), C.c_int64(y.size), C.c_int(mu), _p(out, C.c_double))
    return out


def dilated_index_train(d, dilation):
    """qpnet.py:592-611; float32 d -> tensor path (int64), float64 d -> numpy path (int32)."""
    d = np.ascontiguousarray(d)
    if d.dtype == np.float32:
        out = np.empty(d.shape, dtype=np.int64)
        for b in range(d.shape[0]):
            lib().qpo_dilated_index_train_f32(_p(d[b], C.c_float), C.c_int64(d.shape[1]), C.c_int(dilation), _p(out[b], C.c_int64))
    else:
        d = d.astype(np.float64)
        out = np.empty(d.shape, dtype=np.int32)
        for b in range(d.shape[0]):
            lib().qpo_dilated_index_train_f64(_p(d[b], C.c_double), C.c_int64(d.shape[1]), C.c_int(dilation), _p(out[b], C.c_int32))
    return out


def dilated_index_gen(d, dilation):
    """qpnet.py:613-624."""
    d = np.ascontiguousarray(d)
    if d.dtype == np.float32:
        out = np.empty(d.shape, dtype=np.int64)
        lib().qpo_dilated_index_gen_f32(_p(d, C.c_float), C.c_int64(d.size), C.c_int(dilation), _p(out, C.c_int64))
    else:
        d = d.astype(np.float64)
        out = np.empty(d.shape, dtype=np.int32)
        lib().qpo_dilated_index_gen_f64(_p(d, C.c_double), C.c_int64(d.size), C.c_int(dilation), _p(out, C.c_int32))
    return out


def decode(cfg, flat_w, h, d, x, n_samples, maxd=None, teacher=None, d_is_f32=False,
           want_margin=False, want_logits=False, mode="argmax", seed=0, row=0):
    """One utterance of batch_fast_generate(mode="argmax") (or teacher-forced logits).

    h: (n_aux, F) float32; d: (T,) float64; x: (n_x,) int64 seed.  Returns dict."""
    flat_w = np.ascontiguousarray(flat_w, dtype=np.float32)
    assert flat_w.size == cfg.n_params
    h = np.ascontiguousarray(h, dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.int64)
    if maxd is None:
        maxd = int(np.nanmax(np.ceil(d)))
    out = np.empty(n_samples, dtype=np.int64)
    margin = np.empty(n_samples, dtype=np.float32) if want_margin else None
    logits = np.empty((n_samples, cfg.n_quantize), dtype=np.float32) if want_logits else None
    if teacher is not None:
        teacher = np.ascontiguousarray(teacher, dtype=np.int64)
        assert teacher.size >= n_samples
    rc = lib().qpo_decode(C.byref(_cfg(cfg)), _p(flat_w, C.c_float), _p(h, C.c_float), C.c_int64(h.shape[1]),
                          _p(d, C.c_double), C.c_int64(d.size), _p(x, C.c_int64), C.c_int64(x.size),
                          C.c_int64(n_samples), C.c_int(maxd), _p(teacher, C.c_int64), C.c_int(int(d_is_f32)),
                          _p(out, C.c_int64), _p(margin, C.c_float), _p(logits, C.c_float),
                          C.c_int(1 if mode == "sampling" else 0), C.c_uint64(seed), C.c_int(row))
    if rc != 0:
        raise RuntimeError("qpo_decode failed rc=%d" % rc)
    return {"samples": out, "margin": margin, "logits": logits}


def forward(cfg, flat_w, x, h, d, batch_length):
    """QPNet.forward for ONE batch row (qpnet.py:239-312) restated as teacher-forced streaming:
    x (T,) int64, h (n_aux, F), d (T,) float32 -> logits (batch_length, Q) for the last positions.
    Uses the training index expression (qpnet.py:592-604)."""
    x = np.asarray(x, dtype=np.int64); T = x.size; BL = int(batch_length)
    d32 = np.asarray(d, dtype=np.float32)
    maxd = int(np.ceil(d32).max())
    n_x = T - BL + 1
    teacher = np.concatenate([x[n_x:], [0]])
    r = decode(cfg, flat_w, h, d32.astype(np.float64), x[:n_x], BL, maxd=maxd, teacher=teacher, d_is_f32=2, want_logits=True)
    return r["logits"]


def batch_fast_generate(cfg, flat_w, x, h, n_samples_list, dilated_factors, mode="argmax", seed=0):
    """Batch semantics of QPNet.batch_fast_generate (qpnet.py:314-559): rows are independent,
    padding uses the batch-level ceil(max d); results are returned in completion order
    (ascending length, stable) and `n_samples_list` is consumed the way the reference does."""
    d = np.asarray(dilated_factors)
    d_is_f32 = d.dtype == np.float32
    maxd = int(np.nanmax(np.ceil(d)))
    order = sorted(range(len(n_samples_list)), key=lambda i: n_samples_list[i])
    outs = []
    for i in order:
        r = decode(cfg, flat_w, h[i], d[i].astype(np.float64), x[i], n_samples_list[i], maxd=maxd, d_is_f32=d_is_f32,
                   mode=mode, seed=seed, row=i)
        outs.append(r["samples"])
    keep = n_samples_list[order[-1]]
    del n_samples_list[:]
    n_samples_list.append(keep)
    return outs
