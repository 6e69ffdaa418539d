import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import DEFAULT
from qpnet_amd import synth
import util
cuda = torch.device("cuda:0"); cfg = DEFAULT
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda)
for B in (1, 20):
    bx, bh, bd, ns = synth.decode_batch(cfg, [(100 + b, 10, 1.0) for b in range(B)])
    xb, hb = torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda)
    for G in ("4", "8", "16", "32", "64"):
        os.environ["QPN_DECODE_COOP"] = G
        m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
        print("B=%d cap G=%s: %.1f us/sample/utterance, %.0f samples/s" % (B, G, m.last_decode_kernel_ms * 1e3 / max(ns), sum(ns) / (m.last_decode_kernel_ms * 1e-3)), flush=True)
