# dev: soak test of the stack queues -- many forward + backward passes over chunks of varying length, batch size and workgroup counts, each compared
# with the per-layer launches of the same model (logits bitwise; gradients to float-atomic reassociation).  Races in the hand-offs would show here.
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth, _lib
import util
cuda = torch.device("cuda:0")
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(7)
crit = torch.nn.CrossEntropyLoss()
os.environ["QPN_DROPIN_FLAT_GRAD"] = "0"
W = synth.make_weights(PAPER, 13)
# (the library reads its knobs ONCE per handle, at the first training call: a model per arrangement, made under that arrangement's environment)
_models = {}


def model_for(queue, wgs, wgs_bwd):
    key = (queue, wgs, wgs_bwd)
    if key not in _models:
        os.environ["QPN_STACK_QUEUE"] = "1" if queue else "0"
        os.environ["QPN_STACK_WGS"] = str(wgs); os.environ["QPN_STACK_WGS_BWD"] = str(wgs_bwd)
        mm = util.build_model(PAPER, W, cuda).train()
        hb0 = synth.train_inputs(PAPER, 300, 1, 3000)
        xs = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb0]
        mm(xs[0], xs[1], xs[3], xs[4])                      # the handle's training state (and its knobs) exist from here on
        _models[key] = mm
    return _models[key]


def run(hb, queue, wgs=0, wgs_bwd=0):
    m = model_for(queue, wgs, wgs_bwd)
    x, h, t, d, b = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in hb]
    BL = int(hb[4][0])
    m.zero_grad(set_to_none=True)
    lg = m(x, h, d, b)
    loss = crit(lg.reshape(-1, PAPER.n_quantize), t[:, -BL:].reshape(-1))
    loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    st = (C.c_uint * 16)()
    _lib.lib().qpn_train_stack_stats(m._handle, st, 16, None)
    return lg.detach().clone(), g.clone(), list(st)


worst, bad, esc = 0.0, 0, [0, 0]
t0 = time.time()
for it in range(n_iter):
    full = bool(os.environ.get("SOAK_FULL"))            # the bench shape with the default workgroup counts only
    bl = 20000 if full else int(rs.choice([300, 700, 1500, 4000, 9000, 20000]))
    batch = 2 if (bl <= 4000 and rs.rand() < 0.3) else 1
    hb = synth.train_inputs(PAPER, bl, 1000 + it, 3 * bl + 3000, f0_lo=45.0 if rs.rand() < 0.5 else 70.0, f0_hi=300.0)
    if batch == 2:
        x, h, t, d, b = hb
        xs = rs.randint(0, 256, size=x.shape[1] + 1).astype(np.int64)
        hb = (np.stack([x[0], xs[:-1]]), np.concatenate([h, h]), np.stack([t[0], xs[1:]]), np.concatenate([d, d]), np.concatenate([b, b]))
    wf, wb = (0, 0) if full else (int(rs.choice([128, 256, 384, 512])), int(rs.choice([96, 192, 256, 320, 384, 512])))
    l1, g1, st = run(hb, True, wf, wb)
    l0, g0, _ = run(hb, False)
    same = bool(torch.equal(l1, l0))
    rel = float((g1 - g0).abs().max() / g0.abs().max())
    worst = max(worst, rel); esc[0] += st[4]; esc[1] += st[8]
    if not same or rel > 5e-6 or st[1] != 0:
        bad += 1
        print("MISMATCH it %d bl %d batch %d wgs %s/%s: logits equal %s, grad rel %.2e, control %s" % (it, bl, batch, wf, wb, same, rel, st[:12]), flush=True)
    if it % 20 == 19:
        print("  %d passes, worst gradient difference %.2e, escalations fwd %d bwd %d, %.0f s" % (it + 1, worst, esc[0], esc[1], time.time() - t0), flush=True)
print("soak: %d passes, %d mismatches, worst gradient difference %.2e of the largest gradient" % (n_iter, bad, worst))
sys.exit(1 if bad else 0)
