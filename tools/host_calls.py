# dev: host time of each C-ABI call of one fused step, queue empty in front of the step
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth, _lib
from qpnet_amd.train import FusedTrainer, ensure_flat
import util
cuda = torch.device("cuda:0")
m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
tr = FusedTrainer(m, lr=1e-4)
hb = synth.train_inputs(PAPER, 20000, 5000, 30000, f0_lo=55.0, f0_hi=300.0)
x, h, t, d = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb[:4]]
maxd = int(np.ceil(hb[3]).max()); BL = int(hb[4][0])
for i in range(10): tr.step(x, h, t, d, hb[4], maxd=maxd)
L, hd = m._native(cuda); flat = ensure_flat(m, cuda)
stream = torch.cuda.current_stream(cuda).cuda_stream
print("stream handle", stream)
acc = np.zeros(3)
for i in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(L.qpn_train_forward_loss(hd, flat.data_ptr(), 1, x.shape[1], h.shape[2], d.shape[1], BL, maxd, x.data_ptr(), h.data_ptr(), d.data_ptr(), t.data_ptr(), t.shape[1],
                                        tr._logits.data_ptr(), 0, tr._dlogits.data_ptr(), stream))
    t1 = time.perf_counter()
    _lib.check(L.qpn_train_backward(hd, tr._dlogits.data_ptr(), tr.g.data_ptr(), stream))
    t2 = time.perf_counter()
    tr.step_count += 1
    _lib.check(L.qpn_adam_step_ex(hd, flat.data_ptr(), tr.g.data_ptr(), tr.m.data_ptr(), tr.v.data_ptr(), flat.numel(), tr.step_count, 1e-4, 0.9, 0.999, 1e-8, 0.0, None, stream))
    t3 = time.perf_counter()
    acc += (t1 - t0, t2 - t1, t3 - t2)
print("host ms per call: forward_loss %.3f  backward %.3f  adam %.3f" % tuple(acc / 30 * 1e3))
# the same calls as FusedTrainer.step makes them: wrap the library entry points with timers
import collections
tim = collections.OrderedDict()
class Wrap:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        f = getattr(self._lib, name)
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); tim[name] = tim.get(name, 0.0) + time.perf_counter() - t0; return r
        return g
m._native_cached = None
orig_native = m._native
def native(dev):
    Lr, hdl = orig_native(dev); return Wrap(Lr), hdl
m._native = native
tot = 0.0
for i in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.step(x, h, t, d, hb[4], maxd=maxd); tot += time.perf_counter() - t0
print("FusedTrainer.step: %.3f ms; inside library calls: %s" % (tot / 30 * 1e3, {k: round(v / 30 * 1e3, 3) for k, v in tim.items()}))
