# dev: the one-launch residual stack (train_stack.hip) against the per-layer launches (QPN_STACK_QUEUE=0):
# bitwise comparison of logits / loss / flat gradient, then forward-only and full-step timings of both.
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER, TINY
from qpnet_amd import synth
from qpnet_amd.train import FusedTrainer
import util

cuda = torch.device("cuda:0")


def to(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in arrs]


def run(cfg, bl, seed, ml, queue, batch=1, f0=(55.0, 300.0)):
    os.environ["QPN_STACK_QUEUE"] = "1" if queue else "0"
    m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda).train()
    hb = synth.train_inputs(cfg, bl, seed, ml, f0_lo=f0[0], f0_hi=f0[1])
    if batch == 2:      # second row: same features (rows of a batch share the chunk geometry), another waveform
        x, h, t, d, b = hb
        xs = np.random.RandomState(5).randint(0, cfg.n_quantize, size=x.shape[1] + 1).astype(np.int64)
        hb = (np.stack([x[0], xs[:-1]]), np.concatenate([h, h]), np.stack([t[0], xs[1:]]), np.concatenate([d, d]), np.concatenate([b, b]))
    x, h, t, d = to(*hb[:4])
    b = to(hb[4])[0]
    with torch.no_grad():
        lg = m(x, h, d, b).clone()
    tr = FusedTrainer(m, lr=1e-4)
    losses = [tr.step(x, h, t, d, hb[4], want_loss=True, maxd=int(np.ceil(hb[3]).max())) for _ in range(3)]
    torch.cuda.synchronize()
    return lg.cpu().numpy(), np.array(losses), m._flat.detach().cpu().numpy().copy()


ok = True
for name, cfg, bl, ml in (("tiny", TINY, 500, 3000), ("paper-short", PAPER, 2000, 6000), ("paper-full", PAPER, 20000, 30000)):
    a = run(cfg, bl, 77, ml, True)
    b = run(cfg, bl, 77, ml, False)
    same = [np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y) for x, y in zip(a, b)]
    print(name, "logits / losses / weights-after-3-steps bit-identical:", same, "losses", a[1], b[1], flush=True)
    ok = ok and same[0]
a = run(PAPER, 500, 78, 3000, True, batch=2); b = run(PAPER, 500, 78, 3000, False, batch=2)
print("paper batch 2: logits bit-identical:", np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), flush=True)
ok = ok and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))

# timing: forward only and the full fused step
hbs = [synth.train_inputs(PAPER, 20000, 5000 + 17 * i, 30000, f0_lo=55.0, f0_hi=300.0) for i in range(4)]
bts = [to(*hb[:4]) for hb in hbs]
maxds = [int(np.ceil(hb[3]).max()) for hb in hbs]
for queue in (1, 0, 1, 0):
    os.environ["QPN_STACK_QUEUE"] = str(queue)
    m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    for i in range(30): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("QPN_STACK_QUEUE=%d: %.4f ms/step  %.1f steps/s" % (queue, (t1 - t0) / n * 1e3, n / (t1 - t0)), flush=True)
print("OK" if ok else "MISMATCH")
