# dev: where the host time of the reference-style loop on the drop-in module goes (cProfile, 200 steps)
import sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
import util
cfg = PAPER
cuda = torch.device("cuda:0")
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda).train()
from qpnet_amd.train import FlatAdam
opt = FlatAdam(m, lr=1e-4) if len(sys.argv) > 1 and sys.argv[1] == "flat" else torch.optim.Adam(m.parameters(), lr=1e-4)
crit = torch.nn.CrossEntropyLoss()
batches = []
for i in range(4):
    x, h, t, d, b = synth.train_inputs(cfg, 20000, 5000 + 17 * i, 30000, f0_lo=55.0, f0_hi=300.0)
    batches.append([torch.from_numpy(a).to(cuda) for a in (x, h, t, d, b)])
def step(i):
    x, h, t, d, b = batches[i % 4]
    out = m(x, h, d, b)
    BL = out.shape[1]
    loss = crit(out.view(-1, cfg.n_quantize), t[:, -BL:].reshape(-1))
    opt.zero_grad()
    loss.backward()
    opt.step()
for i in range(10): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(200): step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print("%.1f steps/s under cProfile" % (200 / dt))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
