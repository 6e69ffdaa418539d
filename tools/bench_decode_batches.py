import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
import util
cuda = torch.device("cuda:0")
cfg = PAPER
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for B in [int(a) for a in sys.argv[2:]] or [20, 48, 49, 64, 96, 128]:
    bx, bh, bd, ns = synth.decode_batch(cfg, [(100 + b, F, 1.0) for b in range(B)])
    xb, hb = torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda)
    m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
    t0 = time.perf_counter()
    m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
    dt = time.perf_counter() - t0
    print("B=%d F=%d: %.3f M samples/s, %.2f us/sample/utterance (kernel %.1f ms)  plan: %s" % (B, F, sum(ns) / dt / 1e6, m.last_decode_kernel_ms * 1e3 / max(ns), m.last_decode_kernel_ms, m.last_decode_plan), flush=True)
