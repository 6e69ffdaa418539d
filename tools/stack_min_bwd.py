# dev: ONE forward + backward of a small paper-size chunk, then the stack queue's control words (bisecting a stuck queue; build with -DQPN_STACK_DEBUG)
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth, _lib
import util
cuda = torch.device("cuda:0")
m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
bl = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
hb = synth.train_inputs(PAPER, bl, 77, 3 * bl, f0_lo=55.0, f0_hi=300.0)
x, h, t, d, b = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb]
lg = m(x, h, d, b); bl = int(hb[4][0])
loss = torch.nn.CrossEntropyLoss()(lg.reshape(-1, PAPER.n_quantize), t[:, -bl:].reshape(-1))
try:
    loss.backward()
    torch.cuda.synchronize()
    print("backward ok", float(loss))
except Exception as e:
    print("backward failed:", str(e)[:200])
out = (C.c_uint * 64)()
_lib.lib().qpn_train_stack_stats(m._handle, out, 64, None)
print("control words:", list(out)[:16])
print("bwd debug words (first, n, base, miss lo, miss hi, waiting workgroup, epoch, a missing flag's value):", list(out)[32:40])
