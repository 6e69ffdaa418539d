# dev: is the fused training loop host-bound?  time to ENQUEUE n steps (no synchronisation inside) vs time until the device has finished them
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
hbs = [synth.train_inputs(PAPER, 20000, 5000 + 17 * i, 30000, f0_lo=55.0, f0_hi=300.0) for i in range(4)]
bts = [[torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb[:4]] for hb in hbs]
maxds = [int(np.ceil(hb[3]).max()) for hb in hbs]
for i in range(30): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
torch.cuda.synchronize()
for n in (50, 200, 800):
    t0 = time.perf_counter()
    for i in range(n): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n %4d: enqueue %.3f ms/step, until done %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(400): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# one step at a time with an EMPTY queue in front: the pure enqueue cost (no back-pressure from the runtime's in-flight limits)
ts = []
for i in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4]); ts.append(time.perf_counter() - t0)
print("single step with an empty queue: enqueue median %.3f ms, min %.3f ms" % (float(np.median(ts)) * 1e3, min(ts) * 1e3))
