# dev: the one-launch backward stack (QPN_STACK_QUEUE_BWD, csrc/train_stack.hip) against the per-layer backward launches: flat gradients of
# the same forward (float atomics reassociate: compared relative to the largest gradient), then fused-step timings of both
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth, _lib
from qpnet_amd.train import FusedTrainer
import util
cuda = torch.device("cuda:0")


def to(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in arrs]


def grad(cfg, hb, bq):
    os.environ["QPN_STACK_QUEUE_BWD"] = "1" if bq else "0"
    m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda).train()
    x, h, t, d, b = to(*hb)
    BL = int(hb[4][0])
    lg = m(x, h, d, b)
    loss = torch.nn.CrossEntropyLoss()(lg.reshape(-1, cfg.n_quantize), t[:, -BL:].reshape(-1))
    loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    st = (C.c_uint * 16)()
    _lib.lib().qpn_train_stack_stats(m._handle, st, 16, None)
    return g, float(loss), list(st)[:12]


ok = True
for name, bl, ml, seed, batch in (("paper-short", 2000, 6000, 77, 1), ("paper-full", 20000, 30000, 5000, 1), ("paper batch 2", 700, 3000, 78, 2)):
    hb = synth.train_inputs(PAPER, bl, seed, ml, f0_lo=45.0, f0_hi=300.0)
    if batch == 2:
        x, h, t, d, b = hb
        xs = np.random.RandomState(5).randint(0, 256, size=x.shape[1] + 1).astype(np.int64)
        hb = (np.stack([x[0], xs[:-1]]), np.concatenate([h, h]), np.stack([t[0], xs[1:]]), np.concatenate([d, d]), np.concatenate([b, b]))
    g1, l1, st = grad(PAPER, hb, True)
    g0, l0, _ = grad(PAPER, hb, False)
    g0b, _, _ = grad(PAPER, hb, False)
    rel = np.abs(g1 - g0).max() / np.abs(g0).max()
    noise = np.abs(g0b - g0).max() / np.abs(g0).max()
    print("%s: max |dg| / max |g| = %.2e (two per-layer runs differ by %.2e); control words %s" % (name, rel, noise, st), flush=True)
    ok = ok and rel < 5e-6 and st[1] == 0

hbs = [synth.train_inputs(PAPER, 20000, 5000 + 17 * i, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True) for i in range(4)]
bts = [to(*hb[:4]) for hb in hbs]
maxds = [int(np.ceil(hb[3]).max()) for hb in hbs]
for bq in (1, 0, 1, 0):
    os.environ["QPN_STACK_QUEUE_BWD"] = str(bq)
    m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    for i in range(30): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    st = (C.c_uint * 16)()
    _lib.lib().qpn_train_stack_stats(m._handle, st, 16, None)
    print("QPN_STACK_QUEUE_BWD=%d: %.4f ms/step  %.1f steps/s  control words %s" % (bq, (t1 - t0) / n * 1e3, n / (t1 - t0), list(st)[:12]), flush=True)
print("OK" if ok else "MISMATCH")
