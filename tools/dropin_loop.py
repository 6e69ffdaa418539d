# dev: throughput of the reference-style training loop (model(x,h,d,b) -> nn.CrossEntropyLoss -> backward -> torch.optim.Adam)
# on the drop-in module, next to FusedTrainer (bench.py headline)
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
import util
cfg = PAPER
cuda = torch.device("cuda:0")
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda).train()
from qpnet_amd.train import FlatAdam
opt = FlatAdam(m, lr=1e-4) if len(sys.argv) > 1 else torch.optim.Adam(m.parameters(), lr=1e-4)
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
    return loss
for i in range(5): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 50
for i in range(N): step(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("reference-style loop on the drop-in module: %.1f steps/s (%.3f ms/step)" % (N / dt, dt / N * 1e3))
# serialized breakdown (a synchronize after every stage)
import collections
acc = collections.OrderedDict((k, 0.0) for k in ("forward", "loss", "zero_grad", "backward", "adam"))
def tick():
    torch.cuda.synchronize(); return time.perf_counter()
for i in range(20):
    x, h, t, d, b = batches[i % 4]
    t0 = tick(); out = m(x, h, d, b); t1 = tick()
    loss = crit(out.view(-1, cfg.n_quantize), t[:, -out.shape[1]:].reshape(-1)); t2 = tick()
    opt.zero_grad(); t3 = tick()
    loss.backward(); t4 = tick()
    opt.step(); t5 = tick()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[k] += v
print("serialized ms per stage:", {k: round(v / 20 * 1e3, 3) for k, v in acc.items()})
# host time per stage (no synchronisation inside the loop: what the Python thread spends enqueuing)
acc = collections.OrderedDict((k, 0.0) for k in ("forward", "loss", "zero_grad", "backward", "adam"))
torch.cuda.synchronize()
tl0 = time.perf_counter()
for i in range(100):
    x, h, t, d, b = batches[i % 4]
    t0 = time.perf_counter(); out = m(x, h, d, b); t1 = time.perf_counter()
    loss = crit(out.view(-1, cfg.n_quantize), t[:, -out.shape[1]:].reshape(-1)); t2 = time.perf_counter()
    opt.zero_grad(); t3 = time.perf_counter()
    loss.backward(); t4 = time.perf_counter()
    opt.step(); t5 = time.perf_counter()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[k] += v
tl1 = time.perf_counter()
torch.cuda.synchronize()
tl2 = time.perf_counter()
print("that loop: %.3f ms per step on the host, %.3f ms until the device was done" % ((tl1 - tl0) / 100 * 1e3, (tl2 - tl0) / 100 * 1e3))
print("host ms per stage (unsynchronised loop):", {k: round(v / 100 * 1e3, 3) for k, v in acc.items()})
