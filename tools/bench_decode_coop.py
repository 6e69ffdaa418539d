# dev: throughput of the cooperative decode kernel: default geometry (C=512) and paper-size with QPN_DECODE_COOP
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import DEFAULT, PAPER
from qpnet_amd import synth
import util
cuda = torch.device("cuda:0")
def run(cfg, B, F, tag):
    m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda)
    bx, bh, bd, ns = synth.decode_batch(cfg, [(100 + b, F, 1.0) for b in range(B)])
    xb, hb = torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda)
    m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
    t0 = time.perf_counter()
    m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
    dt = time.perf_counter() - t0
    print("%s B=%d F=%d: %.0f samples/s aggregate, %.1f us/sample/utterance (kernel %.1f ms)" % (tag, B, F, sum(ns) / dt, m.last_decode_kernel_ms * 1e3 / max(ns), m.last_decode_kernel_ms), flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "default"
if which == "default500":                   # the workload bench.py reports for the repo-default geometry: 500-frame (2.5 s) utterances
    for B in (1, 20):
        run(DEFAULT, B, 500, "default (C=512) G=auto")
elif which == "default":
    for B in (1, 4, 8, 20):
        run(DEFAULT, B, 20, "default (C=512) G=auto")
else:
    for G in ("1", "2", "4"):
        os.environ["QPN_DECODE_COOP"] = G
        run(PAPER, 1, 100, "paper coop G=" + G)
    os.environ.pop("QPN_DECODE_COOP")
    run(PAPER, 1, 100, "paper single-CU kernel")
if which == "pipe":
    os.environ["QPN_DECODE_PIPE"] = "1"
    run(PAPER, 1, 300, "paper pipelined (4 CUs/utterance)")
    run(PAPER, 20, 300, "paper pipelined (4 CUs/utterance)")
    run(PAPER, 64, 300, "paper pipelined (4 CUs/utterance)")
    os.environ["QPN_DECODE_PIPE"] = "0"
    run(PAPER, 1, 300, "paper single-CU kernel")
    run(PAPER, 20, 300, "paper single-CU kernel")
