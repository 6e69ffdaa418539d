# dev: soak test of the pipelined decode kernel -- the same batches decoded over and over, every result compared with the first
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import PAPER
from qpnet_amd import synth
import util
cuda = torch.device("cuda:0"); cfg = PAPER
m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda)
def batch(specs):
    bx, bh, bd, ns = synth.decode_batch(cfg, specs)
    return torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), bd, list(ns)
sets = {"B20 x 600 frames": batch([(100 + b, 600, 1.0) for b in range(20)]),
        "B48 ragged": batch([(100 + b, 40 + (b * 7) % 60, [0.5, 1.0, 1.5][b % 3]) for b in range(48)]),
        "B1 x 900 frames": batch([(7, 900, 0.5)])}
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 120
ref = {}; n = 0
while time.time() < t_end:
    for name, (xb, hb, bd, ns) in sets.items():
        for mode in ("argmax", "sampling"):
            m.sampling_seed = 1234
            outs = m.batch_fast_generate(xb, hb, list(ns), bd, mode=mode)
            key = (name, mode)
            if key not in ref: ref[key] = outs
            else:
                for a, b in zip(ref[key], outs):
                    assert np.array_equal(a, b), "mismatch in %s / %s after %d rounds" % (name, mode, n)
    n += 1
    if n % 5 == 0: print("round", n, "ok", flush=True)
print("soak done:", n, "rounds, all identical")
