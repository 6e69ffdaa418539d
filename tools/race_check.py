# dev: the overlapped step (side stream, early reduction, events without system fence) against the same steps with everything on one
# stream (QPN_TRAIN_SERIAL=1), from the same weights on the same chunks.  The two differ only by the order of float atomics in the
# adaptive layers' scatter (fp32 noise); a race -- a kernel reading what another stream has not written yet -- shows as a jump.
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
from qpnet_amd.train import FusedTrainer, ensure_flat
import util
cuda = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
flat0 = synth.make_weights(PAPER, 13)
ma = util.build_model(PAPER, flat0, cuda).train(); mb = util.build_model(PAPER, flat0, cuda).train()
ta, tb = FusedTrainer(ma, lr=1e-4), FusedTrainer(mb, lr=1e-4)
hbs = [synth.train_inputs(PAPER, 20000, 5000 + 17 * i, 30000, f0_lo=55.0, f0_hi=300.0) for i in range(6)]
bts = [[torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb[:4]] for hb in hbs]
maxds = [int(np.ceil(hb[3]).max()) for hb in hbs]
worst = 0.0
for i in range(steps):
    k = i % 6
    os.environ.pop("QPN_TRAIN_SERIAL", None)
    la = ta.step(*bts[k], hbs[k][4], want_loss=(i % 50 == 49), maxd=maxds[k])
    os.environ["QPN_TRAIN_SERIAL"] = "1"
    lb = tb.step(*bts[k], hbs[k][4], want_loss=(i % 50 == 49), maxd=maxds[k])
    if i % 50 == 49:
        wa, wb = ensure_flat(ma, cuda).detach(), ensure_flat(mb, cuda).detach()
        dmax = float((wa - wb).abs().max()); rel = dmax / float(wb.abs().max())
        worst = max(worst, dmax)
        print("step %4d  loss overlapped %.6f serial %.6f  max |dw| %.3e (rel %.1e)" % (i + 1, la, lb, dmax, rel), flush=True)
        assert abs(la - lb) < 1e-3 and np.isfinite(la)
# Adam's update is lr-sized whatever the gradient's scale, so a single flipped sign moves a weight by 2e-4: a few of those are fp32 noise
# amplified, thousands would be a race.  The bound is loose on purpose.
assert worst < 5e-3, worst
print("ok: %d steps, worst max |dw| %.3e" % (steps, worst))
