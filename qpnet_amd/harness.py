"""Host-side helpers of the callers either side of the hot path.

These mirror, with the same names / argument meaning, the pure helpers the reference's
task scripts use to prepare the model inputs:
  dilated_factor   <- src/bin/qpnet_train.py:147-163, src/bin/qpnet_decode.py:90-106
  batch_f0         <- src/bin/qpnet_train.py:165-179
  receptive_field  <- src/bin/qpnet_train.py:181-198
  extend_time      <- src/utils/utils.py:216-235 (nearest-repeat upsample == np.repeat axis 0)
  validate_length  <- src/bin/qpnet_train.py:119-145
  pad_list         <- src/bin/qpnet_decode.py:73-88 (zero padding to the longest item)
File / HDF5 / wav I/O is out of scope.
"""
import numpy as np


def dilated_factor(batch_f0, fs, dense_factor):
    """d = fs / (f0 * dense_factor); f0 == 0 -> d = 1 (float64, like the reference)."""
    f0s = np.array(batch_f0, copy=True)
    f0s[f0s == 0] = fs / dense_factor
    d = np.ones(f0s.shape) * fs
    d /= f0s
    d /= dense_factor
    assert np.all(d > 0)
    return d


def batch_f0(h, f0_threshold=0):
    f0 = h[:, 1].copy(order="C")
    f0[f0 < f0_threshold] = f0_threshold
    return f0


def receptive_field(receptiveCausal_field, receptiveF_field, receptiveA_field, dilated_factors):
    maxd = np.nanmax(dilated_factors)
    return int(receptiveF_field + receptiveA_field * int(np.ceil(maxd)) + receptiveCausal_field)


def extend_time(feats, upsampling_factor):
    """(T x D) -> (T*U x D), each frame repeated U times (float64 like the reference)."""
    return np.repeat(np.asarray(feats, dtype=np.float64), upsampling_factor, axis=0)


def validate_length(x, y, upsampling_factor=None):
    if upsampling_factor is None:
        n = min(x.shape[0], y.shape[0])
        return x[:n], y[:n]
    if x.shape[0] > y.shape[0] * upsampling_factor:
        x = x[:y.shape[0] * upsampling_factor]
    if x.shape[0] < y.shape[0] * upsampling_factor:
        mod_y = y.shape[0] * upsampling_factor - x.shape[0]
        mod_y_frame = mod_y // upsampling_factor + 1
        y = y[:-mod_y_frame]
        x = x[:y.shape[0] * upsampling_factor]
    assert len(x) == len(y) * upsampling_factor
    return x, y


def pad_list(batch_list, pad_value=0.0):
    maxlen = max(b.shape[0] for b in batch_list)
    out = np.full((len(batch_list), maxlen) + batch_list[0].shape[1:], pad_value,
                  dtype=np.asarray(batch_list[0]).dtype)
    for i, b in enumerate(batch_list):
        out[i, :b.shape[0]] = b
    return out


def train_chunk_geometry(cfg, d_buffer, batch_length=20000, max_length=30000):
    """Chunk sizes of the training generator (src/bin/qpnet_train.py:268-284).

    Returns (receptive_field, batch_length_current, h_bs, x_bs)."""
    rf = receptive_field(cfg.receptiveCausal_field, cfg.receptiveF_field, cfg.receptiveA_field, d_buffer)
    mod1 = max(rf + batch_length - max_length, 0)
    bl = batch_length - mod1
    bl -= (rf + bl) % cfg.upsampling_factor
    h_bs = (rf + bl) // cfg.upsampling_factor
    x_bs = h_bs * cfg.upsampling_factor + 1
    return rf, bl, h_bs, x_bs
