# dev: timing of the repo-default geometry (C=512, 12F+4A) training step on the GEMM path, per kernel group
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import DEFAULT, PAPER
from qpnet_amd import synth, _lib
from qpnet_amd.train import FusedTrainer
import util, bench
cfg = DEFAULT if (len(sys.argv) < 2 or sys.argv[1] != "paper") else PAPER
cuda = torch.device("cuda:0")
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda).train()
tr = FusedTrainer(m, lr=1e-4)
hb = [synth.train_inputs(cfg, 20000, 5000 + 17 * i, 30000, f0_lo=55.0, f0_hi=300.0) for i in range(2)]
bt = [[torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in b] for b in hb]
maxds = [int(np.ceil(b[3]).max()) for b in hb]
def step(i, want_loss=False):
    x, h, t, d, bl = bt[i % 2]
    return tr.step(x, h, t, d, hb[i % 2][4], want_loss=want_loss, maxd=maxds[i % 2])
print("loss", step(0, True)); step(1)
torch.cuda.synchronize(); t0 = time.perf_counter(); N = 10
for i in range(N): step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
L_, hd = m._native(cuda)
stream = torch.cuda.current_stream(cuda).cuda_stream
ms = (C.c_float * 10)()
_lib.check(L_.qpn_train_profile_begin(hd, stream))
for i in range(2): step(i)
_lib.check(L_.qpn_train_profile_end(hd, ms, 10, stream))
ms = [v / 2 for v in ms]
x0, h0, t0_, d0, b0 = hb[0]
BL = int(b0[0]); maxd = maxds[0]; N1 = cfg.receptive_field(maxd) + BL - 1
starts, s_ = [], 0
for dil in cfg.dilationsF: s_ += dil; starts.append(s_)
for dil in cfg.dilationsA: s_ += dil * maxd; starts.append(s_)
fl = bench.train_flops(cfg, N1, BL, starts)
print("step %.2f ms = %.1f steps/s; %.2f TFLOP/step -> %.1f TFLOP/s (%.3f of fp32 MFMA peak)" % (dt * 1e3, 1 / dt, sum(fl) / 1e12, sum(fl) / dt / 1e12, sum(fl) / dt / 1e12 / 157.3))
for n, v, f in zip(bench.PG_NAMES, ms, fl + [0]):
    print("  %-12s %8.3f ms  %7.1f TFLOP/s" % (n, v, f / (v * 1e-3) / 1e12 if v > 0 else 0))
