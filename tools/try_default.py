# dev probe: does the repo-"default" geometry (n_resch=512, 12 fixed + 4 adaptive layers) run, and does it match the oracle?
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import DEFAULT
from qpnet_amd import synth
from oracle import cpu_oracle, train_oracle
import util
cfg = DEFAULT
cuda = torch.device("cuda:0")
w = synth.make_weights(cfg, 7)
m = util.build_model(cfg, w, cuda)
F = 4
x, h, d, n = synth.decode_inputs(cfg, F, 11)
try:
    t0 = time.time()
    y = m.batch_fast_generate(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), [n], d[None], mode="argmax")
    print("decode ok %.2fs" % (time.time() - t0), len(y[0]))
    t0 = time.time()
    ref = cpu_oracle.decode(cfg, w, h, d, x, n)["samples"]
    print("oracle %.2fs; match:" % (time.time() - t0), np.array_equal(np.asarray(y[0]), ref))
except Exception as e:
    print("DECODE ERR", repr(e))
try:
    xt, ht, tt, dt, bl = synth.train_inputs(cfg, 1500, 5, max_length=8000)
    out = m(torch.from_numpy(xt).to(cuda), torch.from_numpy(ht).to(cuda), torch.from_numpy(dt).to(cuda), torch.from_numpy(bl))
    print("forward ok", tuple(out.shape))
    ref = train_oracle.forward(cfg, w, xt, ht, dt, int(bl[0])) if hasattr(train_oracle, "forward") else None
    if ref is not None:
        lg = ref[0] if isinstance(ref, tuple) else ref
        print("max abs diff vs numpy oracle", float(np.abs(out.detach().cpu().numpy() - lg).max()))
    loss = out.float().logsumexp(-1).mean()
    loss.backward()
    print("backward ok", float(loss))
except Exception as e:
    print("TRAIN ERR", repr(e))
