# dev: where the run_train loop's time goes.  The same generator chunks (a) pre-staged on the device and stepped in a plain loop (what the device can do
# on THIS chunk mix), (b) the same with the lagged loss, (c) through the real pipeline (generator -> PinnedStager -> prefetch thread -> step):
#   python tools/runner_prof.py
import os, sys, time
if os.environ.get('SWITCH'): sys.setswitchinterval(float(os.environ['SWITCH']))
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd import loaders, synth
from qpnet_amd.config import PAPER
from qpnet_amd.runners import PinnedStager, Prefetcher
from qpnet_amd.train import FusedTrainer
import util
cfg = PAPER
dev = torch.device("cuda:0")
m = util.build_model(cfg, synth.make_weights(cfg, 13), dev).train()
tr = FusedTrainer(m, lr=1e-4)
U = cfg.upsampling_factor
rs = np.random.RandomState(0)
utts = []
for i in range(24):
    nf = int(rs.randint(600, 1200))
    utts.append((rs.uniform(-1, 1, nf * U + 5).astype(np.float32), synth.make_features(nf, 400 + i, 45.0, 300.0)))
mean, scale = synth.scaler_stats()


def gen():
    np.random.seed(1)
    return loaders.train_generator(utts, cfg.receptiveCausal_field, cfg.receptiveF_field, cfg.receptiveA_field, 22050,
                                   wav_transform=loaders.mu_law_transform(cfg.n_quantize), feat_transform=lambda h: (h - mean) / scale,
                                   batch_length=20000, max_length=30000, upsampling_factor=U, shuffle=True)


N = 300
g = gen()
t0 = time.perf_counter()
host = [next(g) for _ in range(N)]
print("generator alone: %.0f chunks/s" % (N / (time.perf_counter() - t0)))
staged = [[(a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev) for a in (bx, bh, bt, bd)] + [bb, int(np.ceil(float(bd.max())))] for bx, bh, bt, bd, bb in host]
print("chunk lengths: min %d max %d" % (min(s[0].shape[1] for s in staged), max(s[0].shape[1] for s in staged)))
for mode in (False, "lagged"):
    for s in staged[:10]:
        tr.step(s[0], s[1], s[2], s[3], s[4], want_loss=mode, maxd=s[5])
    tr.flush_loss(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in staged:
        tr.step(s[0], s[1], s[2], s[3], s[4], want_loss=mode, maxd=s[5])
    th = time.perf_counter() - t0
    tr.flush_loss(); torch.cuda.synchronize()
    print("pre-staged chunks, want_loss=%s: %.0f steps/s (host finished enqueueing after %.0f %% of the wall time)" % (mode, N / (time.perf_counter() - t0), 100 * th / (time.perf_counter() - t0)))

stage = PinnedStager(dev)


def batches():
    for bx, bh, bt, bd, bb in gen():
        dv = stage({"x": bx, "h": bh, "t": bt, "d": bd})
        yield dv["x"], dv["h"], dv["t"], dv["d"], bb, int(np.ceil(float(bd.max())))
stream = Prefetcher(batches())
for _ in range(10):
    b = next(stream); tr.step(*b[:5], want_loss="lagged", maxd=b[5])
tr.flush_loss(); torch.cuda.synchronize(); t0 = time.perf_counter()
tw = ts = 0.0
for i in range(N):
    a = time.perf_counter(); b = next(stream); c = time.perf_counter()
    tr.step(*b[:5], want_loss="lagged", maxd=b[5]); d = time.perf_counter()
    tw += c - a; ts += d - c
tr.flush_loss(); torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("pipeline: %.0f steps/s; per step: waiting for the prefetch thread %.3f ms, inside step() %.3f ms, wall %.3f ms" % (N / wall, tw / N * 1e3, ts / N * 1e3, wall / N * 1e3))
