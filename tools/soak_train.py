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
losses = []
t0 = time.time()
NSTEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for i in range(NSTEPS):
    k = i % 4
    l = tr.step(*bts[k], hbs[k][4], want_loss=(i % (NSTEPS // 12) == 0), maxd=maxds[k])
    if l is not None: losses.append(l); print(i, round(l, 4), flush=True)
torch.cuda.synchronize()
print("%d steps in %.1f s; losses %s" % (NSTEPS, time.time() - t0, [round(x, 3) for x in losses]))
assert all(np.isfinite(losses)) and losses[-1] < losses[0]
