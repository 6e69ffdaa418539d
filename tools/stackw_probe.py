"""dev: device time of the two stack launches ALONE (serial per-launch profile, qpn_train_profile_begin) on the bench chunk, for the launch plan the environment selects
(QPN_STACK_WAVE[_FWD|_BWD], QPN_STACK_WAVES, QPN_STACK_WGS[_BWD], QPN_LIB=<a -DSW_EXP variant>), plus the queues' counters.
    python tools/stackw_probe.py [label]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import util
from qpnet_amd import synth, _lib
from qpnet_amd.config import PAPER
from qpnet_amd.train import FusedTrainer
dev = torch.device("cuda:0")
m = util.build_model(PAPER, synth.make_weights(PAPER, 13), dev).train()
tr = FusedTrainer(m, lr=1e-4)
hbs = [synth.train_inputs(PAPER, 20000, 5000 + 17 * i, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True) for i in range(2)]
bts = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in hb[:4]] for hb in hbs]
def step(i): tr.step(*bts[i % 2], hbs[i % 2][4], want_loss=False, maxd=62)
for i in range(60): step(i)
torch.cuda.synchronize()
L, hd = m._native(dev)
s = torch.cuda.current_stream(dev).cuda_stream
N = 14
ms = (C.c_float * N)()
n = 10
_lib.check(L.qpn_train_profile_begin(hd, s))
for i in range(n): step(i)
_lib.check(L.qpn_train_profile_end(hd, ms, N, s))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
st = (C.c_uint * 16)()
L.qpn_train_stack_stats(hd, st, 16, None)
env = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("QPN_STACK") or k == "QPN_LIB")
print("%-28s stack fwd %.1f us, stack bwd %.1f us alone; two-stream step %.4f ms (%.0f steps/s); bwd queue polls %d  [%s]" % (
    sys.argv[1] if len(sys.argv) > 1 else "", ms[1] / n * 1e3, ms[6] / n * 1e3, dt * 1e3, 1 / dt, st[9], env))
