# dev: decode throughput of argmax vs sampling at batch 1 / 20 (paper-size, 600 frames)
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
import util
cfg = PAPER
cuda = torch.device("cuda:0")
m = util.build_model(cfg, synth.make_weights(cfg, 7), cuda)
for B in (1, 20):
    x, h, d, ns = synth.decode_batch(cfg, [(100 + i, 600, 1.0) for i in range(B)])
    xt, ht = torch.from_numpy(x).to(cuda), torch.from_numpy(h).to(cuda)
    for mode in ("argmax", "sampling"):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            y = m.batch_fast_generate(xt, ht, list(ns), d, mode=mode)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("B=%d %-8s %.0f samples/s" % (B, mode, sum(ns) / dt))
