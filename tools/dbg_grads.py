# dev: per-parameter gradient error of the drop-in module's autograd backward against the numpy oracle (paper size) -- which weight
# gradient a kernel change broke:  python tools/dbg_grads.py [wseed dseed bl]
import numpy as np, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd import synth
from qpnet_amd.config import PAPER
import util
from oracle import train_oracle as TO
cfg=PAPER
cuda=torch.device('cuda:0')
wseed, dseed, bl = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (13, 52, 1200)
flat = synth.make_weights(cfg, wseed)
m = util.build_model(cfg, flat, cuda).train()
x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, 30000)
xt, ht, tt, dt = [torch.from_numpy(a).to(cuda) for a in (x,h,t,d)]
bt=b
BL=int(b[0])
logits = m(xt, ht, dt, bt)
loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
loss.backward()
grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
lg, caches = TO.forward(cfg, flat, x, h, d, b)
print("logits max err %.3e" % np.abs(logits.detach().cpu().numpy() - lg).max())
_, dl = TO.ce_loss(lg, t[:, -BL:])
og = TO.backward(cfg, flat, caches, dl)
offs,_=cfg.param_offsets()
scale=np.abs(og).max()
for k,(o,shp) in offs.items():
    n=int(np.prod(shp)); a,r=grad[o:o+n],og[o:o+n]
    e=np.abs(a-r)
    bound=2e-5*scale+1e-4*np.abs(r).max()
    print("%-36s err %.3e (p99.9 %.2e) ref %.3e bound %.2e %s %s"%(k,e.max(),np.quantile(e,0.999),np.abs(r).max(),bound,"FAIL" if e.max()>bound else "", "ZERO" if np.abs(a).max()==0 else ""))
