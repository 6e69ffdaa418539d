# dev: forward-only loop of the paper-size chunk (for rocprofv3 --kernel-trace --stats) + the stack queue's counters
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth, _lib
from qpnet_amd.train import FusedTrainer
import util
cuda = torch.device("cuda:0")
m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
tr = FusedTrainer(m, lr=1e-4)
hb = synth.train_inputs(PAPER, 20000, 5000, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True)
bt = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb[:4]]
maxd = int(np.ceil(hb[3]).max())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for i in range(n):
    tr.step(*bt, hb[4], want_loss=False, maxd=maxd)
torch.cuda.synchronize()
out = (C.c_uint * 1024)()
_lib.check(_lib.lib().qpn_train_stack_stats(m._handle, out, 1024, None))
print("stack queue control words:", list(out)[:16])
st = np.array(list(out)[600:800], dtype=np.int64).reshape(25, 8)
if st.any():
    print("stamps of workgroup 5 (cycles since its first stamp; columns: top, after barrier 1, gate MFMAs issued, flags checked, epilogue done, after barrier 2, residual phase done, rows staged):")
    for r in st:
        print(" ".join("%7d" % ((int(v) - int(st[0, 0])) & 0xffffffff) if v else "      -" for v in r))
