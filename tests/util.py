"""Helpers shared by the parity tests (inputs regenerated from seeds via qpnet_amd.synth)."""
import numpy as np
from qpnet_amd import synth, harness


def decode_batch(cfg, utts):
    """Prepare a decode batch the way decode_generator does (reference bin/qpnet_decode.py:152-209):
    returns x (B,1) int64, h (B,A,Fmax) f32 zero padded, d (B,Tmax) f64 zero padded, n_samples list."""
    xs, hs, ds, ns = [], [], [], []
    for (fs, nf, fac) in utts:
        x, h, d, n = synth.decode_inputs(cfg, nf, fs, fac)
        xs.append(x); hs.append(h.T); ds.append(d[:, None]); ns.append(n)
    bx = np.stack(xs)
    bh = np.ascontiguousarray(harness.pad_list(hs).transpose(0, 2, 1)).astype(np.float32)
    bd = harness.pad_list(ds).squeeze(-1)
    return bx, bh, bd, ns


def build_model(cfg, flat, device):
    import torch
    from qpnet_amd.qpnet import QPNet
    m = QPNet(**cfg.kwargs())
    sd = {k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()}
    m.load_state_dict(sd)
    return m.to(device).eval()
