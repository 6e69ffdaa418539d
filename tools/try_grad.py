import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import QPNetConfig
from qpnet_amd import synth
from oracle import train_oracle as TO
import util
cuda = torch.device("cuda:0")
C, S = int(sys.argv[1]), int(sys.argv[2])
cfg = QPNetConfig(n_resch=C, n_skipch=S, dilationF_depth=int(sys.argv[3]), dilationF_repeat=1, dilationA_depth=int(sys.argv[4]), dilationA_repeat=1)
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
sc = np.abs(og).max()
for k, (o, shp) in offs.items():
    n = int(np.prod(shp)); e = np.abs(g[o:o+n] - og[o:o+n]).max() / sc
    worst = max(globals().get("worst", 0.0), e)
print("C=%d S=%d F=%s A=%s worst rel err %.2e" % (C, S, sys.argv[3], sys.argv[4], worst))
for k in [] and ["skipF_1x1.0.bias", "skipF_1x1.1.bias", "skipA_1x1.0.bias", "skipA_1x1.1.bias", "resA_1x1.0.bias", "resF_1x1.0.bias"]:
    o, shp = offs[k]
    print(k, "gpu", g[o:o+4], "oracle", og[o:o+4])
print("BL", BL, "x", xt.shape, "maxd", np.ceil(dt).max(), "N1?", cfg.receptive_field(int(np.ceil(dt).max())))
