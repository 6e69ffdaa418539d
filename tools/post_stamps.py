# dev: stage timeline of the post-net forward kernels (a -DQPN_POST_STAMPS build: python qpnet_amd/csrc/build.py --variant poststamps -DQPN_POST_STAMPS;
# QPN_LIB=build_variants/libqpnet_poststamps.so [QPN_POST_HALF=0|1] python tools/post_stamps.py)
import sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd import synth, _lib
from qpnet_amd.config import PAPER
from qpnet_amd.train import FusedTrainer
import util
cfg = PAPER
dev = torch.device("cuda:0")
m = util.build_model(cfg, synth.make_weights(cfg, 13), dev).train()
tr = FusedTrainer(m, lr=1e-4)
x, h, t, d, b = synth.train_inputs(cfg, 20000, 5000, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True)
xs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, h, t, d)]
maxd = int(np.ceil(d).max())
for i in range(5):
    tr.step(*xs, b, want_loss=False, maxd=maxd)
torch.cuda.synchronize()
st = (C.c_uint * 1024)()
_lib.check(_lib.lib().qpn_train_stack_stats(m._handle, st, 1024, None))
names = ["start", "skip-sum done", "put+barrier", "rows_out S0 issued", "gemm P1 done", "put+barrier", "rows_out Y0 issued", "gemm P2 done", "put+barrier", "end (logits, CE)"]
names_b = ["start", "prologue (staging / masks; fused: none)", "gemm dY0 done", "put+barrier", "rows_out dY0 issued", "gemm dS0 done", "put+barrier", "rows_out dS0 issued", "gemms DGS done (2 x 256 columns)", "end (outputs stored from the accumulators)"]
for slot, what in ((0, "forward tile, wg 5"), (1, "backward tile, wg 5")):
    if slot == 1: names = names_b
    v = [st[600 + 16 * slot + i] for i in range(10)]
    if v[0] == 0 and v[9] == 0:
        continue
    print(what)
    for i in range(10):
        print("  %-22s +%7d cycles  (total %7d)" % (names[i], (v[i] - v[i - 1]) & 0xffffffff if i else 0, (v[i] - v[0]) & 0xffffffff))
