"""Training-side plumbing over the C ABI: autograd.Function for QPNet.forward and a fused
train step (forward + CE + backward + Adam) that never leaves libqpnet_hip for compute.

torch is used for device buffers, streams and the autograd graph only."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def ensure_flat(model, dev):
    """One contiguous fp32 buffer holding every parameter in state_dict order, with the
    nn.Parameters re-pointed at views of it (so optimizers / load_state_dict write into it and
    the C ABI reads it without a gather).  Rebuilt if .to()/.cuda() replaced the storages."""
    params = list(model.parameters())
    flat = getattr(model, "_flat", None)
    ok = flat is not None and flat.device == dev
    if ok:
        base = flat.data_ptr()
        o = 0
        for p in params:
            if p.data_ptr() != base + 4 * o or p.dtype != torch.float32:
                ok = False
                break
            o += p.numel()
    if not ok:
        flat = torch.cat([p.detach().reshape(-1).to(dev, torch.float32) for p in params]).contiguous()
        o = 0
        for p in params:
            n = p.numel()
            p.data = flat[o:o + n].view(p.shape)
            o += n
        model._flat = flat
        model._gflat = None
    return model._flat


def _split_like(model, gflat):
    out, o = [], 0
    for p in model.parameters():
        n = p.numel()
        out.append(gflat[o:o + n].view(p.shape))
        o += n
    return out


class QPNetFunction(torch.autograd.Function):
    """logits = QPNet.forward(x, h, d, blength); backward = hand-written HIP kernels."""

    @staticmethod
    def forward(ctx, model, x, h, d, BL, maxd, *params):
        dev = x.device
        L, hd = model._native(dev)
        flat = ensure_flat(model, dev)
        B, T = x.shape
        logits = torch.empty((B, BL, model.n_quantize), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(L.qpn_train_forward(hd, flat.data_ptr(), B, T, h.shape[2], d.shape[1], BL, maxd,
                                           x.data_ptr(), h.data_ptr(), d.data_ptr(), logits.data_ptr(), stream))
            _lib.check(L.qpn_train_status(hd, stream))       # reference asserts on the gather bounds (qpnet.py:294)
        ctx.model = model
        ctx.keep = (x, h, d)                                  # inputs must outlive backward (the workspace points into them)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        dev = dlogits.device
        L, hd = model._native(dev)
        flat = model._flat
        if getattr(model, "_gflat", None) is None:
            model._gflat = torch.empty_like(flat)
        g = model._gflat
        dl = dlogits.contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(L.qpn_train_backward(hd, dl.data_ptr(), g.data_ptr(), stream))
        grads = _split_like(model, g)
        return (None, None, None, None, None, None) + tuple(grads)


def qpnet_forward(model, x, h, dilated_factors, blength):
    """QPNet.forward (reference qpnet.py:239-262 for the argument handling)."""
    dev = x.device
    if dev.type != "cuda":
        raise RuntimeError("qpnet_amd.QPNet runs on an AMD GPU only (tensors are on %s); there is no CPU fallback" % dev)
    assert torch.all(blength == blength[0])
    BL = int(blength[0])
    maxd = int(torch.max(dilated_factors.ceil()))
    x = x.to(dev, torch.int64).contiguous()
    h = h.to(dev, torch.float32).contiguous()
    d = dilated_factors.to(dev, torch.float32).contiguous()
    ensure_flat(model, dev)
    return QPNetFunction.apply(model, x, h, d, BL, maxd, *list(model.parameters()))


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (lr, betas, eps, weight_decay; no amsgrad) as ONE kernel over the model's flat parameter
    buffer -- a one-line swap for `torch.optim.Adam(model.parameters(), lr=...)` in the reference loop
    (src/bin/qpnet_train.py:426-429,531), whose ~50-tensor foreach update costs 0.9 ms per step on this GPU.
    Gradients are read from `p.grad` (the autograd backward hands out views of one flat gradient buffer, so no gather
    happens on the usual path).  State (`exp_avg`, `exp_avg_sq`, `step`) lives in flat tensors; its state_dict is its own
    format, not torch.optim.Adam's."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._m = self._v = None
        self._steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        model = self.model
        flat = getattr(model, "_flat", None)
        if flat is None:
            raise RuntimeError("FlatAdam.step before the first forward/backward of the model")
        dev = flat.device
        L, hd = model._native(dev)
        params = list(model.parameters())
        g = getattr(model, "_gflat", None)
        ok = g is not None
        if ok:
            o = 0
            for p in params:
                if p.grad is None or p.grad.data_ptr() != g.data_ptr() + 4 * o:
                    ok = False
                    break
                o += p.numel()
        if not ok:                                   # grads came from elsewhere (or some are None): gather them
            g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
        if self._m is None or self._m.numel() != flat.numel() or self._m.device != dev:
            self._m = torch.zeros_like(flat); self._v = torch.zeros_like(flat)
        grp = self.param_groups[0]
        self._steps += 1
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(L.qpn_adam_step(hd, flat.data_ptr(), g.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), flat.numel(),
                                       self._steps, grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"], stream))
        return loss

    def state_dict(self):
        return {"flat_adam": {"exp_avg": self._m, "exp_avg_sq": self._v, "step": self._steps}, "param_groups": [
            {k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        st = sd["flat_adam"]
        self._m, self._v, self._steps = st["exp_avg"], st["exp_avg_sq"], int(st["step"])
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v


class FusedTrainer:
    """forward + mean CE + backward + Adam entirely behind the C ABI (flat fp32 buffers).

    Mirrors the loop body of the reference trainer (src/bin/qpnet_train.py:517-531) with
    torch.optim.Adam(lr=1e-4, betas=(0.9,0.999), eps=1e-8, weight_decay=0) semantics.
    With world_size > 1 the flat gradient is all-reduced (one RCCL call) before the update."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None, world_size=1):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.step_count = 0
        self.m = self.v = None
        self.pg, self.world = process_group, world_size

    def step(self, x, h, t, d, blength, want_loss=True, maxd=None):
        """One optimisation step.  maxd = ceil(max(d)) of the chunk; a loader that built d on the host (the reference's
        train_generator does, qpnet_train.py:268-272) passes it in, otherwise it is read back from the device (one sync)."""
        model = self.model
        dev = x.device
        L, hd = model._native(dev)
        flat = ensure_flat(model, dev)
        if self.m is None:
            self.m = torch.zeros_like(flat); self.v = torch.zeros_like(flat)
            self.g = torch.empty_like(flat)
        BL = int(blength[0])
        if maxd is None:
            maxd = int(torch.max(d.ceil()))
        B, T = x.shape
        if getattr(self, "_logits", None) is None or self._logits.shape != (B, BL, model.n_quantize):
            self._logits = torch.empty((B, BL, model.n_quantize), dtype=torch.float32, device=dev)
            self._dlogits = torch.empty_like(self._logits)
        stream = torch.cuda.current_stream(dev).cuda_stream
        loss = C.c_double(0.0)
        with torch.cuda.device(dev):
            _lib.check(L.qpn_train_forward(hd, flat.data_ptr(), B, T, h.shape[2], d.shape[1], BL, maxd,
                                           x.data_ptr(), h.data_ptr(), d.data_ptr(), self._logits.data_ptr(), stream))
            _lib.check(L.qpn_ce_loss(hd, self._logits.data_ptr(), t.data_ptr(), t.shape[1], B, BL,
                                     self._dlogits.data_ptr(), C.byref(loss) if want_loss else None, stream))
            _lib.check(L.qpn_train_backward(hd, self._dlogits.data_ptr(), self.g.data_ptr(), stream))
            if self.world > 1:
                from .parallel import allreduce_mean_gradient
                allreduce_mean_gradient(self.g, B * BL, group=self.pg)
            self.step_count += 1
            _lib.check(L.qpn_adam_step(hd, flat.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), flat.numel(),
                                       self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, stream))
        return loss.value if want_loss else None
