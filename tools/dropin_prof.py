# dev: cProfile of the reference-style loop's Python side on the drop-in module
import sys, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
import util
cfg = PAPER
cuda = torch.device("cuda:0")
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
crit = torch.nn.CrossEntropyLoss()
x, h, t, d, b = [torch.from_numpy(a).to(cuda) for a in synth.train_inputs(cfg, 20000, 5000, 30000, f0_lo=55.0, f0_hi=300.0)]
def step():
    out = m(x, h, d, b)
    loss = crit(out.view(-1, cfg.n_quantize), t[:, -out.shape[1]:].reshape(-1))
    opt.zero_grad()
    loss.backward()
    opt.step()
import time
for _ in range(20): step()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): step()
    torch.cuda.synchronize()
    print("reference-style loop: %.1f steps/s" % (300 / (time.perf_counter() - t0)), flush=True)
if "--rate" in sys.argv:
    sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
