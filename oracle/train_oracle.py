"""numpy ORACLE of the QPNet training step (forward + CE + backward + Adam).  TEST INFRASTRUCTURE ONLY.

A from-scratch float32 restatement (time-major, hand-derived backward) of
  QPNet.forward ............................. reference src/nets/qpnet.py:239-312
  _fixed/_adaptive_residual_forward ......... reference src/nets/qpnet.py:626-670
  _dilated_index (tensor path) .............. reference src/nets/qpnet.py:592-604
  CE(mean) + Adam(lr 1e-4, betas .9/.999) ... reference src/bin/qpnet_train.py:426-430,526-531
Pinned by tests/golden/train.npz (loss per step, grads of step 0 and final weights produced by
the imported reference + torch autograd).  Never imported by the product package.

`with precision(np.float64):` runs the SAME code with float64 arithmetic (weights, activations, contractions; the tap positions keep the
reference's float32 rounding, which is part of the algorithm: qpnet.py:592-604) -- the yardstick tests/test_train_gpu.py measures both the
float32 oracle and the GPU gradient against, so that the full-size tolerances are a measured multiple of fp32 reassociation error.
"""
import contextlib

import numpy as np

f32 = np.float32            # the working precision (see precision())
_idx32 = np.float32         # the tap arithmetic's precision: fixed by the reference


@contextlib.contextmanager
def precision(dtype):
    global f32
    old, f32 = f32, dtype
    try:
        yield
    finally:
        f32 = old


def unpack(cfg, flat):
    offs, total = cfg.param_offsets()
    assert flat.size == total
    return {k: flat[o:o + int(np.prod(s))].reshape(s) for k, (o, s) in offs.items()}


def _layers(cfg):
    out = [("F", i, dil) for i, dil in enumerate(cfg.dilationsF)]
    out += [("A", i, dil) for i, dil in enumerate(cfg.dilationsA)]
    return out


def _sigmoid(z):
    return (1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(f32)


def dilated_index(d_tail, dil):
    L = d_tail.shape[0]
    s = (-(d_tail.astype(_idx32)) * _idx32(dil)).astype(_idx32) + np.arange(-L, 0).astype(_idx32)
    return np.rint(s.astype(_idx32)).astype(np.int64)          # negative index from the end of the layer input


def forward_row(cfg, W, x, h, d, BL, maxd):
    """one batch row. x (T,) int, h (A,F) f32, d (T,) f32 -> logits (BL,Q), cache"""
    C, S, Q, A, U = cfg.n_resch, cfg.n_skipch, cfg.n_quantize, cfg.n_aux, cfg.upsampling_factor
    recF, recA = cfg.receptiveF_field, cfg.receptiveA_field * maxd
    N0 = recA + recF + 1 + BL
    xs = (x[-N0:] % Q).astype(np.int64)
    cw, cb = W["causal.conv.weight"], W["causal.conv.bias"]
    X = (cw[:, xs[:-1], 0] + cw[:, xs[1:], 1]).T.astype(f32) + cb          # (N1, C)
    if U > 0:
        uw, ub = W["upsampling.conv.weight"].reshape(-1), W["upsampling.conv.bias"][0]
        hup = (h[:, :, None] * uw[None, None, :] + ub).reshape(A, -1).T.astype(f32)   # (T, A)
    else:
        hup = h.T.astype(f32)
    cache = {"xs": xs, "hup": hup, "layers": [], "h": h}
    skips = []
    for kind, i, dil in _layers(cfg):
        Lin = X.shape[0]
        if kind == "F":
            shift = dil
            xc, xp = X[shift:], X[:-shift]
            ws, wt = W["dilF_sigmoid.%d.conv.weight" % i], W["dilF_tanh.%d.conv.weight" % i]
            Wcs, Wps, Wct, Wpt = ws[:, :, 1], ws[:, :, 0], wt[:, :, 1], wt[:, :, 0]
            bs_, bt_ = W["dilF_sigmoid.%d.conv.bias" % i], W["dilF_tanh.%d.conv.bias" % i]
            idx = None
        else:
            shift = dil * maxd
            Lout = Lin - shift
            idx = dilated_index(d[-Lout:], dil)
            assert -idx.min() <= Lin
            xc, xp = X[shift:], X[Lin + idx]
            Wcs, Wps = W["dilA_sigmoid.%d.convC.weight" % i][:, :, 0], W["dilA_sigmoid.%d.convP.weight" % i][:, :, 0]
            Wct, Wpt = W["dilA_tanh.%d.convC.weight" % i][:, :, 0], W["dilA_tanh.%d.convP.weight" % i][:, :, 0]
            bs_ = W["dilA_sigmoid.%d.convC.bias" % i] + W["dilA_sigmoid.%d.convP.bias" % i]
            bt_ = W["dilA_tanh.%d.convC.bias" % i] + W["dilA_tanh.%d.convP.bias" % i]
        Lout = Lin - shift
        hh = hup[-Lout:]
        Vs, Vt = W["aux%s_1x1_sigmoid.%d.weight" % (kind, i)][:, :, 0], W["aux%s_1x1_tanh.%d.weight" % (kind, i)][:, :, 0]
        vbs, vbt = W["aux%s_1x1_sigmoid.%d.bias" % (kind, i)], W["aux%s_1x1_tanh.%d.bias" % (kind, i)]
        zs = xc @ Wcs.T + xp @ Wps.T + hh @ Vs.T + (bs_ + vbs)
        zt = xc @ Wct.T + xp @ Wpt.T + hh @ Vt.T + (bt_ + vbt)
        sg, th = _sigmoid(zs), np.tanh(zt.astype(np.float64)).astype(f32)
        g = sg * th
        Wr, br = W["res%s_1x1.%d.weight" % (kind, i)][:, :, 0], W["res%s_1x1.%d.bias" % (kind, i)]
        Wk, bk = W["skip%s_1x1.%d.weight" % (kind, i)][:, :, 0], W["skip%s_1x1.%d.bias" % (kind, i)]
        out = g @ Wr.T + br + xc
        skips.append(g[-BL:] @ Wk.T + bk)
        cache["layers"].append(dict(kind=kind, i=i, shift=shift, idx=idx, xc=xc, xp=xp, hh=hh, sg=sg, th=th, g=g, Lin=Lin))
        X = out
    s0 = np.sum(skips, axis=0).astype(f32)
    s1 = np.maximum(s0, 0)
    y0 = s1 @ W["conv_post_1.weight"][:, :, 0].T + W["conv_post_1.bias"]
    y1 = np.maximum(y0, 0)
    logits = y1 @ W["conv_post_2.weight"][:, :, 0].T + W["conv_post_2.bias"]
    cache.update(s0=s0, s1=s1, y0=y0, y1=y1, BL=BL, N1=N0 - 1)
    return logits.astype(f32), cache


def forward(cfg, flat, x, h, d, blength):
    W = unpack(cfg, flat)
    BL = int(blength[0])
    maxd = int(np.ceil(d).max())
    outs, caches = [], []
    for b in range(x.shape[0]):
        lg, c = forward_row(cfg, W, x[b], h[b], d[b], BL, maxd)
        outs.append(lg); caches.append(c)
    return np.stack(outs), caches


def ce_loss(logits, targets):
    """mean CE over all rows (torch.nn.CrossEntropyLoss default) -> loss, dlogits"""
    lg = logits.reshape(-1, logits.shape[-1]).astype(np.float64)
    t = targets.reshape(-1)
    m = lg.max(1, keepdims=True)
    e = np.exp(lg - m)
    lse = np.log(e.sum(1)) + m[:, 0]
    loss = float((lse - lg[np.arange(t.size), t]).mean())
    p = e / e.sum(1, keepdims=True)
    p[np.arange(t.size), t] -= 1.0
    return loss, (p / t.size).astype(f32).reshape(logits.shape)


def backward_row(cfg, W, G, c, dlogits):
    """accumulate parameter grads of one batch row into dict G (same keys/shapes as W)."""
    C, S, Q, A, U = cfg.n_resch, cfg.n_skipch, cfg.n_quantize, cfg.n_aux, cfg.upsampling_factor
    BL = c["BL"]
    W2, W1 = W["conv_post_2.weight"][:, :, 0], W["conv_post_1.weight"][:, :, 0]
    G["conv_post_2.weight"][:, :, 0] += dlogits.T @ c["y1"]; G["conv_post_2.bias"] += dlogits.sum(0)
    dy0 = (dlogits @ W2) * (c["y0"] > 0)
    G["conv_post_1.weight"][:, :, 0] += dy0.T @ c["s1"]; G["conv_post_1.bias"] += dy0.sum(0)
    ds0 = ((dy0 @ W1) * (c["s0"] > 0)).astype(f32)                       # (BL,S) grad of every layer's skip
    dhup = np.zeros_like(c["hup"])
    dX = None                                                           # grad wrt the current layer's OUTPUT
    for (kind, i, dil), lc in reversed(list(zip(_layers(cfg), c["layers"]))):
        Lout = lc["g"].shape[0]
        g, sg, th = lc["g"], lc["sg"], lc["th"]
        Wr = W["res%s_1x1.%d.weight" % (kind, i)][:, :, 0]
        Wk = W["skip%s_1x1.%d.weight" % (kind, i)][:, :, 0]
        dg = np.zeros((Lout, C), f32)
        dg[-BL:] += ds0 @ Wk
        G["skip%s_1x1.%d.weight" % (kind, i)][:, :, 0] += ds0.T @ g[-BL:]; G["skip%s_1x1.%d.bias" % (kind, i)] += ds0.sum(0)
        if dX is not None:
            dg += dX @ Wr
            G["res%s_1x1.%d.weight" % (kind, i)][:, :, 0] += dX.T @ g; G["res%s_1x1.%d.bias" % (kind, i)] += dX.sum(0)
        dzs = dg * th * sg * (1 - sg)
        dzt = dg * sg * (1 - th * th)
        xc, xp, hh = lc["xc"], lc["xp"], lc["hh"]
        G["aux%s_1x1_sigmoid.%d.weight" % (kind, i)][:, :, 0] += dzs.T @ hh; G["aux%s_1x1_sigmoid.%d.bias" % (kind, i)] += dzs.sum(0)
        G["aux%s_1x1_tanh.%d.weight" % (kind, i)][:, :, 0] += dzt.T @ hh; G["aux%s_1x1_tanh.%d.bias" % (kind, i)] += dzt.sum(0)
        Vs, Vt = W["aux%s_1x1_sigmoid.%d.weight" % (kind, i)][:, :, 0], W["aux%s_1x1_tanh.%d.weight" % (kind, i)][:, :, 0]
        dhup[-Lout:] += dzs @ Vs + dzt @ Vt
        dXin = np.zeros((lc["Lin"], C), f32)
        if kind == "F":
            ws, wt = W["dilF_sigmoid.%d.conv.weight" % i], W["dilF_tanh.%d.conv.weight" % i]
            gs, gt = G["dilF_sigmoid.%d.conv.weight" % i], G["dilF_tanh.%d.conv.weight" % i]
            gs[:, :, 1] += dzs.T @ xc; gs[:, :, 0] += dzs.T @ xp; gt[:, :, 1] += dzt.T @ xc; gt[:, :, 0] += dzt.T @ xp
            G["dilF_sigmoid.%d.conv.bias" % i] += dzs.sum(0); G["dilF_tanh.%d.conv.bias" % i] += dzt.sum(0)
            dxc = dzs @ ws[:, :, 1] + dzt @ wt[:, :, 1]
            dxp = dzs @ ws[:, :, 0] + dzt @ wt[:, :, 0]
            dXin[lc["shift"]:] += dxc
            dXin[:-lc["shift"]] += dxp
        else:
            for half, dz in (("sigmoid", dzs), ("tanh", dzt)):
                G["dilA_%s.%d.convC.weight" % (half, i)][:, :, 0] += dz.T @ xc; G["dilA_%s.%d.convC.bias" % (half, i)] += dz.sum(0)
                G["dilA_%s.%d.convP.weight" % (half, i)][:, :, 0] += dz.T @ xp; G["dilA_%s.%d.convP.bias" % (half, i)] += dz.sum(0)
            dxc = dzs @ W["dilA_sigmoid.%d.convC.weight" % i][:, :, 0] + dzt @ W["dilA_tanh.%d.convC.weight" % i][:, :, 0]
            dxp = dzs @ W["dilA_sigmoid.%d.convP.weight" % i][:, :, 0] + dzt @ W["dilA_tanh.%d.convP.weight" % i][:, :, 0]
            dXin[lc["shift"]:] += dxc
            np.add.at(dXin, lc["Lin"] + lc["idx"], dxp)                 # scatter-add (collisions possible)
        if dX is not None:
            dXin[lc["shift"]:] += dX                                    # residual connection
        dX = dXin
    # causal conv = two table lookups
    xs = c["xs"]
    gw = G["causal.conv.weight"]
    np.add.at(gw[:, :, 0].T, xs[:-1], dX)
    np.add.at(gw[:, :, 1].T, xs[1:], dX)
    G["causal.conv.bias"] += dX.sum(0)
    if U > 0:
        h = c["h"]
        F = h.shape[1]
        dh3 = dhup.T.reshape(A, F, U)
        G["upsampling.conv.weight"].reshape(-1)[:] += np.einsum("afu,af->u", dh3, h)
        G["upsampling.conv.bias"] += dhup.sum()


def backward(cfg, flat, caches, dlogits):
    W = unpack(cfg, flat)
    gflat = np.zeros_like(flat)
    G = unpack(cfg, gflat)
    for b, c in enumerate(caches):
        backward_row(cfg, W, G, c, dlogits[b])
    return gflat


class Adam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) in float32."""

    def __init__(self, n, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
        self.m = np.zeros(n, f32); self.v = np.zeros(n, f32); self.t = 0
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps

    def step(self, flat, grad):
        self.t += 1
        self.m += (grad - self.m) * f32(1 - self.b1)
        self.v = self.v * f32(self.b2) + f32(1 - self.b2) * grad * grad
        bc1 = 1 - self.b1 ** self.t; bc2 = 1 - self.b2 ** self.t
        denom = np.sqrt(self.v) / f32(np.sqrt(bc2)) + f32(self.eps)
        flat -= f32(self.lr / bc1) * self.m / denom
        return flat


def train_step(cfg, flat, opt, x, h, t, d, b):
    logits, caches = forward(cfg, flat, x, h, d, b)
    BL = int(b[0])
    loss, dlogits = ce_loss(logits, t[:, -BL:])
    grad = backward(cfg, flat, caches, dlogits)
    opt.step(flat, grad)
    return loss, grad
