# dev: per-parameter gradient error of the drop-in module's autograd backward against the numpy oracle (paper size) -- which weight
# gradient a kernel change broke:  python tools/dbg_grads.py
import numpy as np, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd import synth
from qpnet_amd.config import PAPER
import util
from oracle import train_oracle as TO
cfg=PAPER
cuda=torch.device('cuda:0')
flat = synth.make_weights(cfg, 3)
m = util.build_model(cfg, flat, cuda).train()
x, h, t, d, b = synth.train_inputs(cfg, 2000, 5, 30000)
xt, ht, tt, dt = [torch.from_numpy(a).to(cuda) for a in (x,h,t,d)]
bt=b
BL=int(b[0])
logits = m(xt, ht, dt, bt)
loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
loss.backward()
grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
lg, caches = TO.forward(cfg, flat, x, h, d, b)
_, dl = TO.ce_loss(lg, t[:, -BL:])
og = TO.backward(cfg, flat, caches, dl)
offs,_=cfg.param_offsets()
for k,(o,shp) in offs.items():
    n=int(np.prod(shp)); a,r=grad[o:o+n],og[o:o+n]
    print("%-28s err %.3e ref %.3e %s"%(k,np.abs(a-r).max(),np.abs(r).max(), "ZERO" if np.abs(a).max()==0 else ""))
