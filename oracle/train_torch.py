"""torch-CPU ORACLE of the QPNet training step (forward + CE + autograd backward + torch.optim.Adam).  TEST INFRASTRUCTURE ONLY.

The same time-major restatement as oracle/train_oracle.py (which is pinned to the reference by tests/golden/train.npz), written with
torch float32 ops so that the elementwise work, the gathers and the contractions all use the host's cores the way the reference's own
CPU path does -- bench.py's `cpu_baseline` leg times THIS (kind "port"); the numpy oracle stays the checker of the GPU tests.
  QPNet.forward ............................. reference src/nets/qpnet.py:239-312
  _fixed/_adaptive_residual_forward ......... reference src/nets/qpnet.py:626-670
  _dilated_index (tensor path) .............. reference src/nets/qpnet.py:592-604
  CE(mean) + Adam(lr 1e-4, betas .9/.999) ... reference src/bin/qpnet_train.py:426-430,526-531
tests/test_oracle_golden.py pins it to the same reference fixture as the numpy oracle (losses, the full step-0 gradient, the weights after
the Adam steps: tests/golden/train.npz).  Never imported by the product package.
"""
import numpy as np
import torch


def _views(cfg, flat):
    offs, total = cfg.param_offsets()
    assert flat.numel() == total
    return {k: flat[o:o + int(np.prod(s))].view(*s) for k, (o, s) in offs.items()}


def _layers(cfg):
    return [("F", i, dil) for i, dil in enumerate(cfg.dilationsF)] + [("A", i, dil) for i, dil in enumerate(cfg.dilationsA)]


def _dilated_index(d_tail, dil):
    # float32 arithmetic in the reference's order: rint(f32(-d * dil) + f32(idx))   (qpnet.py:592-604)
    L = d_tail.shape[0]
    s = (-(d_tail.to(torch.float32)) * float(dil)) + torch.arange(-L, 0, dtype=torch.float32)
    return torch.round(s).to(torch.int64)


def forward_row(cfg, W, x, h, d, BL, maxd):
    """one batch row: x (T,) int64, h (A,F) f32, d (T,) f32 -> logits (BL,Q)"""
    C, Q, A, U = cfg.n_resch, cfg.n_quantize, cfg.n_aux, cfg.upsampling_factor
    recF, recA = cfg.receptiveF_field, cfg.receptiveA_field * maxd
    N0 = recA + recF + 1 + BL
    xs = torch.remainder(x[-N0:], Q)
    cw, cb = W["causal.conv.weight"], W["causal.conv.bias"]
    X = (cw[:, xs[:-1], 0] + cw[:, xs[1:], 1]).t() + cb                      # (N1, C)
    if U > 0:
        uw, ub = W["upsampling.conv.weight"].reshape(-1), W["upsampling.conv.bias"][0]
        hup = (h[:, :, None] * uw[None, None, :] + ub).reshape(A, -1).t()     # (T, A)
    else:
        hup = h.t()
    skips = []
    for kind, i, dil in _layers(cfg):
        Lin = X.shape[0]
        if kind == "F":
            shift = dil
            xc, xp = X[shift:], X[:-shift]
            ws, wt = W["dilF_sigmoid.%d.conv.weight" % i], W["dilF_tanh.%d.conv.weight" % i]
            Wcs, Wps, Wct, Wpt = ws[:, :, 1], ws[:, :, 0], wt[:, :, 1], wt[:, :, 0]
            bs_, bt_ = W["dilF_sigmoid.%d.conv.bias" % i], W["dilF_tanh.%d.conv.bias" % i]
        else:
            shift = dil * maxd
            idx = _dilated_index(d[-(Lin - shift):], dil)
            assert int(-idx.min()) <= Lin
            xc, xp = X[shift:], X[Lin + idx]
            Wcs, Wps = W["dilA_sigmoid.%d.convC.weight" % i][:, :, 0], W["dilA_sigmoid.%d.convP.weight" % i][:, :, 0]
            Wct, Wpt = W["dilA_tanh.%d.convC.weight" % i][:, :, 0], W["dilA_tanh.%d.convP.weight" % i][:, :, 0]
            bs_ = W["dilA_sigmoid.%d.convC.bias" % i] + W["dilA_sigmoid.%d.convP.bias" % i]
            bt_ = W["dilA_tanh.%d.convC.bias" % i] + W["dilA_tanh.%d.convP.bias" % i]
        hh = hup[-(Lin - shift):]
        Vs, Vt = W["aux%s_1x1_sigmoid.%d.weight" % (kind, i)][:, :, 0], W["aux%s_1x1_tanh.%d.weight" % (kind, i)][:, :, 0]
        vbs, vbt = W["aux%s_1x1_sigmoid.%d.bias" % (kind, i)], W["aux%s_1x1_tanh.%d.bias" % (kind, i)]
        zs = xc @ Wcs.t() + xp @ Wps.t() + hh @ Vs.t() + (bs_ + vbs)
        zt = xc @ Wct.t() + xp @ Wpt.t() + hh @ Vt.t() + (bt_ + vbt)
        g = torch.sigmoid(zs) * torch.tanh(zt)
        Wr, br = W["res%s_1x1.%d.weight" % (kind, i)][:, :, 0], W["res%s_1x1.%d.bias" % (kind, i)]
        Wk, bk = W["skip%s_1x1.%d.weight" % (kind, i)][:, :, 0], W["skip%s_1x1.%d.bias" % (kind, i)]
        skips.append(g[-BL:] @ Wk.t() + bk)
        X = g @ Wr.t() + br + xc
    s1 = torch.relu(torch.stack(skips).sum(0))
    y1 = torch.relu(s1 @ W["conv_post_1.weight"][:, :, 0].t() + W["conv_post_1.bias"])
    return y1 @ W["conv_post_2.weight"][:, :, 0].t() + W["conv_post_2.bias"]


class Trainer:
    """flat float32 parameters + torch.optim.Adam, one `step` = the reference trainer's loop body (qpnet_train.py:517-531)."""

    def __init__(self, cfg, flat, lr=1e-4):
        self.cfg = cfg
        self.flat = torch.tensor(np.asarray(flat, np.float32), requires_grad=True)
        self.opt = torch.optim.Adam([self.flat], lr=lr)

    def loss_and_grad(self, x, h, t, d, blength):
        cfg = self.cfg
        W = _views(cfg, self.flat)
        BL = int(np.asarray(blength).reshape(-1)[0])
        maxd = int(np.ceil(np.asarray(d)).max())
        xt, ht, dt = torch.as_tensor(np.asarray(x, np.int64)), torch.as_tensor(np.asarray(h, np.float32)), torch.as_tensor(np.asarray(d, np.float32))
        logits = torch.stack([forward_row(cfg, W, xt[b], ht[b], dt[b], BL, maxd) for b in range(xt.shape[0])])
        tt = torch.as_tensor(np.asarray(t, np.int64))[:, -BL:]
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, cfg.n_quantize), tt.reshape(-1))
        self.opt.zero_grad()
        loss.backward()
        return float(loss.detach()), self.flat.grad.detach().numpy()

    def step(self, x, h, t, d, blength):
        loss, _ = self.loss_and_grad(x, h, t, d, blength)
        self.opt.step()
        return loss
