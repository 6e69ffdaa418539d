"""Synthetic WORLD-shaped inputs and seeded weights.

No corpus, checkpoint or network is available, so tests, smoke() and bench.py use
synthetic features of VCC2018 shape: per utterance F frames of
``[uv, cont_f0 (Hz), 35 mcep, 2 codeap]`` (layout: reference src/bin/feature_extract.py:337-343),
f0 a smooth positive contour inside the corpus range 45-450 Hz
(corpus/VCC2018/conf/pow_f0_dict.yml), the other dims ~N(0,1).
``np.random.RandomState`` is used on purpose: its streams are frozen across numpy versions,
so golden fixtures only need to store seeds + expected outputs.
"""
import numpy as np
from .config import QPNetConfig
from . import harness

FS = 22050
DENSE_FACTOR = 8
F0_MEAN, F0_SCALE = 150.0, 60.0     # identity scaler except the f0 column


def make_weights(cfg: QPNetConfig, seed: int = 1, gain: float = 1.0) -> np.ndarray:
    """Flat fp32 parameter vector in state_dict order; Xavier-uniform conv weights,
    small random biases, upsampling kernel ~1 (so w[j], b are exercised)."""
    rs = np.random.RandomState(seed)
    offs, total = cfg.param_offsets()
    flat = np.zeros(total, dtype=np.float32)
    for key, (o, shp) in offs.items():
        n = int(np.prod(shp))
        if key.startswith("upsampling"):
            v = (1.0 + 0.1 * rs.standard_normal(n)) if key.endswith("weight") else 0.05 * rs.standard_normal(n)
        elif key.endswith("bias"):
            v = 0.05 * rs.standard_normal(n)
        else:
            fan_out = shp[0] * (shp[2] if len(shp) > 2 else 1)
            fan_in = shp[1] * (shp[2] if len(shp) > 2 else 1)
            a = gain * np.sqrt(6.0 / (fan_in + fan_out))
            v = rs.uniform(-a, a, n)
        flat[o:o + n] = v.astype(np.float32)
    return flat


def weights_to_state_dict(cfg: QPNetConfig, flat: np.ndarray):
    offs, _ = cfg.param_offsets()
    return {k: flat[o:o + int(np.prod(s))].reshape(s).copy() for k, (o, s) in offs.items()}


def make_features(n_frames: int, seed: int = 1, f0_lo: float = 80.0, f0_hi: float = 300.0,
                  n_aux: int = 39) -> np.ndarray:
    """Raw (un-normalised) WORLD-like features (F, n_aux) float32."""
    rs = np.random.RandomState(seed)
    # smooth f0: low-passed random walk in log-f0, mapped into [lo, hi]
    w = rs.standard_normal(n_frames + 64)
    k = np.hanning(41); k /= k.sum()
    walk = np.convolve(np.cumsum(w) * 0.15, k, mode="same")[32:32 + n_frames]
    z = 0.5 + 0.5 * np.sin(walk)                       # in [0,1], smooth
    f0 = f0_lo * (f0_hi / f0_lo) ** z
    uv = np.ones(n_frames)
    # unvoiced runs >= 20 frames, ~30 % of frames
    t = 0
    while t < n_frames:
        run = int(rs.randint(20, 80))
        if rs.rand() < 0.3:
            uv[t:t + run] = 0.0
        t += run
    h = rs.standard_normal((n_frames, n_aux))
    h[:, 0] = uv
    h[:, 1] = f0
    return h.astype(np.float32)


def scaler_stats(n_aux: int = 39):
    mean = np.zeros(n_aux); scale = np.ones(n_aux)
    mean[1], scale[1] = F0_MEAN, F0_SCALE
    return mean, scale


def decode_inputs(cfg: QPNetConfig, n_frames: int, seed: int = 1, f0_factor: float = 1.0,
                  f0_lo: float = 80.0, f0_hi: float = 300.0):
    """Inputs of ``batch_fast_generate`` for ONE utterance, prepared the way
    ``decode_generator`` does (reference src/bin/qpnet_decode.py:164-200):
    returns x (1,) int64 seed, h (n_aux, F) float32 normalised, d (F*U,) float64, n_samples."""
    h = make_features(n_frames, seed, f0_lo, f0_hi, cfg.n_aux)
    h[:, 1] = h[:, 1] * np.float32(f0_factor)
    d = harness.dilated_factor(harness.batch_f0(h), FS, DENSE_FACTOR)
    d = harness.extend_time(d[:, None], cfg.upsampling_factor)[:, 0]
    mean, scale = scaler_stats(cfg.n_aux)
    hn = ((h - mean) / scale).astype(np.float32)
    x = np.array([cfg.n_quantize // 2], dtype=np.int64)   # encode_mu_law(0) = 128
    n_samples = n_frames * cfg.upsampling_factor - 1
    return x, np.ascontiguousarray(hn.T), d, n_samples


def train_inputs(cfg: QPNetConfig, batch_length: int, seed: int = 1, max_length: int = 30000,
                 f0_lo: float = 80.0, f0_hi: float = 300.0, batch_size: int = 1, pin_f0_floor: bool = False):
    """One training batch shaped like ``train_generator`` yields it
    (reference src/bin/qpnet_train.py:250-327): x (B,T) int64, h (B,n_aux,F) f32,
    t (B,T) int64 targets, d (B,T) f32, blength (B,) int64."""
    U = cfg.upsampling_factor
    xs, hs, ts, ds, bs = [], [], [], [], []
    # enough frames for one chunk at the worst-case receptive field
    n_frames = (max_length + batch_length) // U + 8
    for b in range(batch_size):
        h = make_features(n_frames, seed + 1000 * b, f0_lo, f0_hi, cfg.n_aux)
        if pin_f0_floor:      # the lowest pitch of the chunk IS the stated floor, so ceil(max d) -- and with it the receptive field -- is the stated one
            h[int(np.argmin(h[:min(n_frames, 150), 1])), 1] = np.float32(f0_lo)
        d = harness.dilated_factor(harness.batch_f0(h, 0), FS, DENSE_FACTOR)
        d = harness.extend_time(d[:, None], U)[:, 0].astype(np.float32)
        rf, bl, h_bs, x_bs = harness.train_chunk_geometry(cfg, d, batch_length, max_length)
        rs = np.random.RandomState(seed + 7 + 1000 * b)
        x_ = rs.randint(0, cfg.n_quantize, size=x_bs).astype(np.int64)
        mean, scale = scaler_stats(cfg.n_aux)
        h_ = ((h[:h_bs] - mean) / scale).astype(np.float32)
        xs.append(x_[:-1]); ts.append(x_[1:]); hs.append(h_.T.copy()); ds.append(d[:x_bs][:-1]); bs.append(bl)
    assert len(set(bs)) == 1, "all rows of a batch must share batch_length (qpnet.py:253)"
    return (np.stack(xs), np.stack(hs), np.stack(ts), np.stack(ds), np.array(bs, dtype=np.int64))


def decode_batch(cfg: QPNetConfig, utts):
    """A decode batch prepared the way ``decode_generator`` does (reference bin/qpnet_decode.py:152-209).
    utts: [(feature seed, n_frames, f0_factor)].  Returns x (B,1) int64, h (B,A,Fmax) f32 zero padded,
    d (B,Tmax) f64 zero padded, n_samples list."""
    xs, hs, ds, ns = [], [], [], []
    for (fs, nf, fac) in utts:
        x, h, d, n = decode_inputs(cfg, nf, fs, fac)
        xs.append(x); hs.append(h.T); ds.append(d[:, None]); ns.append(n)
    bx = np.stack(xs)
    bh = np.ascontiguousarray(harness.pad_list(hs).transpose(0, 2, 1)).astype(np.float32)
    bd = harness.pad_list(ds).squeeze(-1)
    return bx, bh, bd, ns
