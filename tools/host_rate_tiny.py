# dev: host cost of one fused step's launches -- the TINY geometry has almost no device work, so enqueue time per step is host time
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import TINY, PAPER
from qpnet_amd import synth
from qpnet_amd.train import FusedTrainer
import util
cuda = torch.device("cuda:0")
for name, cfg, bl in (("tiny", TINY, 200), ("paper, 2000-sample chunks", PAPER, 2000)):
    m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    hb = synth.train_inputs(cfg, bl, 5000, 30000)
    bt = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb[:4]]
    maxd = int(np.ceil(hb[3]).max())
    for i in range(30): tr.step(*bt, hb[4], want_loss=False, maxd=maxd)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): tr.step(*bt, hb[4], want_loss=False, maxd=maxd)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: enqueue %.3f ms/step, until done %.3f ms/step" % (name, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
