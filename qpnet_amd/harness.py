"""Host-side helpers of the callers either side of the hot path.

Same names / argument meaning / results as the pure helpers the reference's task scripts use to
prepare the model inputs (behaviour pinned by the KATs in tests/test_host_logic.py):
  dilated_factor   <- src/bin/qpnet_train.py:147-163, src/bin/qpnet_decode.py:90-106
  batch_f0         <- src/bin/qpnet_train.py:165-179
  receptive_field  <- src/bin/qpnet_train.py:181-198
  extend_time      <- src/utils/utils.py:216-235 (nearest-repeat upsample == np.repeat axis 0)
  validate_length  <- src/bin/qpnet_train.py:119-145
  pad_list         <- src/bin/qpnet_decode.py:73-88 (zero padding to the longest item)
"""
import numpy as np


def dilated_factor(batch_f0, fs, dense_factor):
    """Pitch-dependent dilation d = (fs / f0) / dense_factor in float64; unvoiced frames (f0 == 0) get the
    f0 that makes d == 1.  The two divisions are kept separate (that is the order the results were pinned in)."""
    f0 = np.asarray(batch_f0)
    unvoiced_f0 = f0.dtype.type(fs / dense_factor) if f0.dtype.kind == "f" else fs / dense_factor
    f0_eff = np.where(f0 == 0, unvoiced_f0, f0).astype(np.float64)
    d = (np.float64(fs) / f0_eff) / dense_factor
    assert np.all(d > 0)
    return d


def batch_f0(h, f0_threshold=0):
    """f0 column of the feature matrix, floored at f0_threshold."""
    return np.maximum(np.ascontiguousarray(h[:, 1]), h.dtype.type(f0_threshold))


def receptive_field(receptiveCausal_field, receptiveF_field, receptiveA_field, dilated_factors):
    """RF_F + RF_A * ceil(max d) + RF_causal."""
    worst = int(np.ceil(np.nanmax(dilated_factors)))
    return int(receptiveF_field + receptiveA_field * worst + receptiveCausal_field)


def extend_time(feats, upsampling_factor):
    """(T x D) -> (T*U x D), each frame repeated U times (float64 like the reference)."""
    return np.repeat(np.asarray(feats, dtype=np.float64), upsampling_factor, axis=0)


def validate_length(x, y, upsampling_factor=None):
    """Trim a (waveform, frames) pair to consistent lengths.  With an upsampling factor U the result has
    len(x) == len(y) * U; a waveform that is SHORT of its frames costs one frame more than strictly needed
    (shortfall // U + 1 frames are dropped), which is the reference's behaviour."""
    if upsampling_factor is None:
        n = min(x.shape[0], y.shape[0])
        return x[:n], y[:n]
    U = int(upsampling_factor)
    frames = y.shape[0]
    shortfall = frames * U - x.shape[0]
    if shortfall > 0:
        frames = max(frames - (shortfall // U + 1), 0)
    x, y = x[:frames * U], y[:frames]
    assert len(x) == len(y) * U
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
    return (rf,) + chunk_plan(rf, batch_length, max_length, cfg.upsampling_factor)


def chunk_plan(rf, batch_length, max_length, upsampling_factor):
    """(batch_length_current, frames per chunk, samples per chunk): batch_length is cut so that RF + BL fits
    max_length and is a whole number of frames; a chunk carries one extra sample for the input/target shift."""
    bl = batch_length - max(rf + batch_length - max_length, 0)
    bl -= (rf + bl) % upsampling_factor
    frames = (rf + bl) // upsampling_factor
    return bl, frames, frames * upsampling_factor + 1
