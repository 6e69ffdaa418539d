# dev: run_train's loop (bench.runner_loop_rate) under Prefetcher depth / GIL switch-interval settings
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from qpnet_amd.config import PAPER
from qpnet_amd import synth
from qpnet_amd.train import FusedTrainer
import util
cuda = torch.device("cuda:0")
m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
tr = FusedTrainer(m, lr=1e-4)
for si in (None, 0.001, 0.0002):
    if si: sys.setswitchinterval(si)
    for depth in (2, 4):
        os.environ["QPN_PREFETCH_DEPTH"] = str(depth)
        r = [bench.runner_loop_rate(m, tr, PAPER, cuda, steps=300) for _ in range(2)]
        print("switch interval %s depth %d: %.0f %.0f steps/s" % (si, depth, r[0], r[1]), flush=True)
