# dev: is the fused training step limited by the host's enqueue rate?  Enqueue time per step (no sync) vs device time per step.
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
from qpnet_amd.train import FusedTrainer
import util
cuda = torch.device("cuda:0")
m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
tr = FusedTrainer(m, lr=1e-4)
hb = synth.train_inputs(PAPER, 20000, 5000, 30000, f0_lo=55.0, f0_hi=300.0)
x, h, t, d = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb[:4]]
maxd = int(np.ceil(hb[3]).max())
for i in range(20): tr.step(x, h, t, d, hb[4], maxd=maxd)
torch.cuda.synchronize()
for n in (20, 100, 400):
    t0 = time.perf_counter()
    for i in range(n): tr.step(x, h, t, d, hb[4], maxd=maxd)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d steps: host enqueue %.3f ms/step, until the device is done %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
# pure host cost: the same step on a tiny chunk (device work ~0.1 ms)
hb2 = synth.train_inputs(PAPER, 1000, 5000, 3000, f0_lo=55.0, f0_hi=300.0)
x2, h2, t2_, d2 = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb2[:4]]
maxd2 = int(np.ceil(hb2[3]).max())
for i in range(20): tr.step(x2, h2, t2_, d2, hb2[4], maxd=maxd2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(400): tr.step(x2, h2, t2_, d2, hb2[4], maxd=maxd2)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("tiny chunk, 400 steps: host enqueue %.3f ms/step, until the device is done %.3f ms/step" % ((t1 - t0) / 400 * 1e3, (t2 - t0) / 400 * 1e3), flush=True)
