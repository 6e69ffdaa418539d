"""Helpers shared by the parity tests (inputs regenerated from seeds via qpnet_amd.synth)."""
import numpy as np
from qpnet_amd import synth, harness


decode_batch = synth.decode_batch


def build_model(cfg, flat, device):
    import torch
    from qpnet_amd.qpnet import QPNet
    m = QPNet(**cfg.kwargs())
    sd = {k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()}
    m.load_state_dict(sd)
    return m.to(device).eval()
