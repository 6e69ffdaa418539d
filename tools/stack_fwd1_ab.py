# dev: the one-workgroup-per-CU software-pipelined forward (QPN_STACK_FWD1=1) against the default stack queue: bitwise logits, then rates
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth, _lib
from qpnet_amd.train import FusedTrainer
import util
cuda = torch.device("cuda:0")
to = lambda *a: [torch.from_numpy(np.ascontiguousarray(x)).to(cuda) for x in a]
ok = True
for name, bl, ml, seed, batch in (("paper-short", 2000, 6000, 77, 1), ("paper-full", 20000, 30000, 5000, 1), ("paper batch 2", 700, 3000, 78, 2)):
    hb = synth.train_inputs(PAPER, bl, seed, ml, f0_lo=45.0, f0_hi=300.0)
    if batch == 2:
        x, h, t, d, b = hb
        hb = (np.concatenate([x, (x + 7) % 256]), np.concatenate([h, h]), np.concatenate([t, t]), np.concatenate([d, d]), np.concatenate([b, b]))
    x, h, t, d, b = to(*hb)
    outs = []
    for v in ("1", "0"):
        os.environ["QPN_STACK_FWD1"] = v
        m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda)
        with torch.no_grad():
            outs.append(m(x, h, d, b).cpu().numpy())
        st = (C.c_uint * 16)(); _lib.lib().qpn_train_stack_stats(m._handle, st, 16, None)
        if v == "1": st1 = list(st)[:8]
    same = np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    print(name, "logits bit-identical:", same, "control words", st1, flush=True)
    ok = ok and same
hbs = [synth.train_inputs(PAPER, 20000, 5000 + 17 * i, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True) for i in range(4)]
bts = [to(*hb[:4]) for hb in hbs]
maxds = [int(np.ceil(hb[3]).max()) for hb in hbs]
for v in (1, 0, 1, 0):
    os.environ["QPN_STACK_FWD1"] = str(v)
    m = util.build_model(PAPER, synth.make_weights(PAPER, 13), cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    for i in range(30): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
    torch.cuda.synchronize(); n = 300; t0 = time.perf_counter()
    for i in range(n): tr.step(*bts[i % 4], hbs[i % 4][4], want_loss=False, maxd=maxds[i % 4])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = (C.c_uint * 16)(); _lib.lib().qpn_train_stack_stats(m._handle, st, 16, None)
    print("QPN_STACK_FWD1=%d: %.4f ms/step  %.1f steps/s  control words %s" % (v, (t1 - t0) / n * 1e3, n / (t1 - t0), list(st)[:8]), flush=True)
print("OK" if ok else "MISMATCH")
