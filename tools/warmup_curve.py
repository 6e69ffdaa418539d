"""Why are the first steps of a process slower?  Per-step device time (events around each step) and host enqueue time of the first N fused steps
on the bench chunk, in a fresh process; optional pre-spin of the GPU with a dummy torch kernel (clock ramp hypothesis).
    python tools/warmup_curve.py [--steps 80] [--spin-ms 0] [--sync-every 0]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=80)
ap.add_argument("--spin-ms", type=float, default=0.0)
ap.add_argument("--chunks", type=int, default=4)
ap.add_argument("--lr", type=float, default=1e-4)
ap.add_argument("--spin-after-first", type=float, default=0.0, help="ms of a dummy torch matmul loop BETWEEN the first (initialising) step and the rest")
ap.add_argument("--prewarm", type=int, default=0, help="steps on ANOTHER model + trainer (its own native handle) before the measured run")
ap.add_argument("--sleep-ms", type=float, default=0.0, help="idle time between the prewarm and the measured run")
args = ap.parse_args()
import torch
from qpnet_amd import synth
from qpnet_amd.config import PAPER
from qpnet_amd.qpnet import QPNet
from qpnet_amd.train import FusedTrainer
dev = torch.device("cuda:0")
cfg = PAPER
flat = synth.make_weights(cfg, 13)
m = QPNet(**cfg.kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
m = m.to(dev).train()
tr = FusedTrainer(m, lr=args.lr)
hb = [synth.train_inputs(cfg, 20000, 5000 + 17 * i, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True) for i in range(args.chunks)]
bt = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in b] for b in hb]
torch.cuda.synchronize()
if args.spin_ms > 0:
    a = torch.randn(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < args.spin_ms:
        a = a @ a * 1e-4
        torch.cuda.synchronize()
if args.prewarm > 0:
    m0 = QPNet(**cfg.kwargs())
    m0.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
    m0 = m0.to(dev).train()
    tr0 = FusedTrainer(m0, lr=1e-4)
    for i in range(args.prewarm):
        x, h, t, d, _ = bt[i % args.chunks]
        tr0.step(x, h, t, d, hb[i % args.chunks][4], want_loss=False, maxd=62)
    torch.cuda.synchronize()
    if args.sleep_ms > 0:
        time.sleep(args.sleep_ms * 1e-3)
N = args.steps
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
host = []
t_all = time.perf_counter()
for i in range(N):
    x, h, t, d, _ = bt[i % args.chunks]
    t0 = time.perf_counter()
    evs[i][0].record()
    tr.step(x, h, t, d, hb[i % args.chunks][4], want_loss=False, maxd=62)
    evs[i][1].record()
    host.append((time.perf_counter() - t0) * 1e3)
    if i == 0 and args.spin_after_first > 0:
        a = torch.randn(4096, 4096, device=dev)
        ts = time.perf_counter()
        while (time.perf_counter() - ts) * 1e3 < args.spin_after_first:
            a = a @ a * 1e-4
            torch.cuda.synchronize()
        del a
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) * 1e3
devms = [a.elapsed_time(b) for a, b in evs]
gap = [evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(N - 1)]
print("lr %g prewarm %d sleep %.0f |" % (args.lr, args.prewarm, args.sleep_ms), "spin %.0f ms; %d steps wall %.2f ms (%.1f steps/s)" % (args.spin_ms, N, wall, N / wall * 1e3))
for lo in range(0, N, 10):
    hi = min(lo + 10, N)
    print("steps %3d-%3d: device ms/step %s | host enqueue ms %s | gaps %s" % (
        lo, hi - 1, " ".join("%.3f" % v for v in devms[lo:hi]), " ".join("%.2f" % v for v in host[lo:hi]), " ".join("%.3f" % v for v in gap[lo:min(hi, N - 1)])))
# windows like the driver's: warmup 5 + 20 timed, vs later windows
for w0 in (5, 25, 45):
    if w0 + 20 <= N:
        tot = evs[w0][0].elapsed_time(evs[w0 + 19][1])
        print("window steps %d..%d: %.3f ms/step (%.1f steps/s)" % (w0, w0 + 19, tot / 20, 20 / tot * 1e3))
