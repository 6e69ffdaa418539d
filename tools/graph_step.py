# dev: does replaying the training step as a hipGraph shorten it?  (north_star mentions hipGraph)
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
from qpnet_amd.train import FusedTrainer
import util
cfg = PAPER
cuda = torch.device("cuda:0")
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda).train()
tr = FusedTrainer(m, lr=1e-4)
hb = synth.train_inputs(cfg, 20000, 5000, 30000, f0_lo=55.0, f0_hi=300.0)
x, h, t, d, b = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb]
maxd = int(np.ceil(hb[3]).max())
def step(): tr.step(x, h, t, d, hb[4], want_loss=False, maxd=maxd)
for _ in range(5): step()
def timeit(fn, n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager            : %.4f ms/step" % timeit(step))
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    print("hipGraph replay  : %.4f ms/step" % timeit(g.replay))
    print("eager again      : %.4f ms/step" % timeit(step))
except Exception as e:
    print("graph capture failed:", repr(e))
