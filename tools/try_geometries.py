# dev probe: which geometries beyond the BASELINE ones run (decode vs oracle, training forward/backward)?
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qpnet_amd.config import QPNetConfig
from qpnet_amd import synth
from oracle import cpu_oracle, train_oracle as TO
import util
cuda = torch.device("cuda:0")
for C, S, fd, fr, ad, ar in [(128, 128, 2, 1, 2, 1), (64, 128, 3, 2, 2, 1), (16, 64, 2, 1, 1, 1), (96, 256, 2, 1, 1, 1)]:
    cfg = QPNetConfig(n_resch=C, n_skipch=S, dilationF_depth=fd, dilationF_repeat=fr, dilationA_depth=ad, dilationA_repeat=ar)
    tag = "C=%d S=%d F=%dx%d A=%dx%d" % (C, S, fd, fr, ad, ar)
    try:
        w = synth.make_weights(cfg, 7)
        m = util.build_model(cfg, w, cuda)
        x, h, d, n = synth.decode_inputs(cfg, 4, 11)
        y = m.batch_fast_generate(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), [n], d[None], mode="argmax")
        ref = cpu_oracle.decode(cfg, w, h, d, x, n)["samples"]
        print(tag, "decode:", "bit-exact" if np.array_equal(np.asarray(y[0]), ref) else "MISMATCH")
    except Exception as e:
        print(tag, "decode ERR", str(e)[:150])
    try:
        xt, ht, tt, dt, bl = synth.train_inputs(cfg, 600, 5, max_length=4000)
        m = util.build_model(cfg, w, cuda).train()
        out = m(torch.from_numpy(xt).to(cuda), torch.from_numpy(ht).to(cuda), torch.from_numpy(dt).to(cuda), torch.from_numpy(bl))
        lg, caches = TO.forward(cfg, w, xt, ht, dt, bl)
        err = float(np.abs(out.detach().cpu().numpy() - lg).max())
        BL = int(bl[0])
        loss = torch.nn.CrossEntropyLoss()(out.reshape(-1, cfg.n_quantize), torch.from_numpy(tt).to(cuda)[:, -BL:].reshape(-1))
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
        _, dl = TO.ce_loss(lg, tt[:, -BL:]); og = TO.backward(cfg, w, caches, dl)
        print(tag, "train: logits err %.2e grad rel err %.2e" % (err, float(np.abs(g - og).max() / np.abs(og).max())))
    except Exception as e:
        print(tag, "train ERR", str(e)[:150])
