import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import QPNetConfig
from qpnet_amd import synth
from oracle import train_oracle as TO
import util
cuda = torch.device("cuda:0")
cfg = QPNetConfig(n_resch=128, n_skipch=128, dilationF_depth=2, dilationF_repeat=1, dilationA_depth=2, dilationA_repeat=1)
w = synth.make_weights(cfg, 7)
xt, ht, tt, dt, bl = synth.train_inputs(cfg, 600, 5, max_length=4000)
m = util.build_model(cfg, w, cuda).train()
out = m(torch.from_numpy(xt).to(cuda), torch.from_numpy(ht).to(cuda), torch.from_numpy(dt).to(cuda), torch.from_numpy(bl))
lg, caches = TO.forward(cfg, w, xt, ht, dt, bl)
BL = int(bl[0])
loss = torch.nn.CrossEntropyLoss()(out.reshape(-1, cfg.n_quantize), torch.from_numpy(tt).to(cuda)[:, -BL:].reshape(-1))
loss.backward()
g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
_, dl = TO.ce_loss(lg, tt[:, -BL:]); og = TO.backward(cfg, w, caches, dl)
offs, _ = cfg.param_offsets()
for k, (o, shp) in offs.items():
    n = int(np.prod(shp)); a, r = g[o:o+n], og[o:o+n]
    e = np.abs(a - r).max() / (np.abs(r).max() + 1e-30)
    bad = np.nonzero(np.abs(a - r) > 1e-3 * np.abs(r).max())[0]
    print("%-36s %-16s own-rel err %.2e  n_bad %d %s" % (k, shp, e, bad.size, bad[:6]))
