# dev: fused-step rate (paper-size chunk) + the stack queue's counters of the last step
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth, _lib
from qpnet_amd.train import FusedTrainer
import util
cuda = torch.device("cuda:0")
m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
tr = FusedTrainer(m, lr=1e-4)
hbs = [synth.train_inputs(PAPER, 20000, 5000 + 17 * i, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True) for i in range(4)]      # bench.py's shape
bts = [[torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb[:4]] for hb in hbs]
maxds = [int(np.ceil(hb[3]).max()) for hb in hbs]
for i in range(30): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
torch.cuda.synchronize()
n = 400
t0 = time.perf_counter()
for i in range(n): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
torch.cuda.synchronize()
t1 = time.perf_counter()
out = (C.c_uint * 16)()
_lib.lib().qpn_train_stack_stats(m._handle, out, 16, None)
print("%.4f ms/step  %.1f steps/s   queue counters (escalations, polls, late looks) fwd %d %d %d bwd %d %d %d" % ((t1 - t0) / n * 1e3, n / (t1 - t0), out[4], out[5], out[6], out[8], out[9], out[10]))
