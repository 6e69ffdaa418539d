# dev: cooperative decode of the repo-default geometry (C = 512) at B = 1 and B = 20 for several group sizes G (QPN_DECODE_COOP caps G; a fresh module per setting):
#   python tools/coop_group_sweep.py [G ...]
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd import synth
from qpnet_amd.config import DEFAULT
import util
cfg = DEFAULT
dev = torch.device("cuda:0")
flat = synth.make_weights(cfg, 13)
FR = 200
for G in (sys.argv[1:] or ["64", "128", "256"]):
    os.environ["QPN_DECODE_COOP"] = G
    for B in (1, 20):
        m = util.build_model(cfg, flat, dev)
        bx, bh, bd, ns = synth.decode_batch(cfg, [(100 + b, FR, 1.0) for b in range(B)])
        xb, hb = torch.from_numpy(bx).to(dev), torch.from_numpy(bh).to(dev)
        wx, wh, wd, wn = synth.decode_batch(cfg, [(100, 10, 1.0)])
        m.batch_fast_generate(torch.from_numpy(wx).to(dev), torch.from_numpy(wh).to(dev), list(wn), wd, mode="argmax")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("G<=%s B=%d: %.1f k samples/s, %.1f us per sample per utterance (%s)" % (G, B, sum(ns) / dt / 1e3, m.last_decode_kernel_ms * 1e3 / max(ns), m.last_decode_plan), flush=True)
