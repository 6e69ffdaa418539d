"""Training-side plumbing over the C ABI: autograd.Function for QPNet.forward and a fused
train step (forward + CE + backward + Adam) that never leaves libqpnet_hip for compute.

torch is used for device buffers, streams and the autograd graph only.

One outstanding forward per model: the activations backward needs live in the native handle's
workspace (sized for 288 GB of HBM: whole-chunk activations stay resident), so a second forward of
the same model before the first one's backward would replace them.  Every forward gets a generation
number from the library; a backward whose forward is no longer the current one raises instead of
using the wrong activations."""
import os
import weakref
import ctypes as C

import numpy as np
import torch

from . import _lib


# Walking a module tree for its parameters costs ~0.15 ms of Python per call (120 tensors, 60 sub-modules), and the reference-style
# loop needs the list three times per step.  The list is cached per model and dropped whenever ANY parameter is registered on any module
# (torch's global registration hook: `m.x = nn.Parameter(...)`, `register_parameter`); storages replaced by .to()/.cuda() keep the
# Parameter objects and are caught by ensure_flat's pointer check.
_PARAM_GENERATION = [0]


def _on_parameter_registered(module, name, param):
    _PARAM_GENERATION[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_on_parameter_registered)
# ... and whenever a sub-module is registered (`model.x = other_module` brings parameters without registering any)
torch.nn.modules.module.register_module_module_registration_hook(_on_parameter_registered)


class _AdoptedAdam:
    """What the step hooks keep per stock `torch.optim.Adam` they step themselves (see _adam_prehook)."""
    __slots__ = ("model", "n", "m", "v", "steps", "step_t", "stash", "mviews", "vviews")


_EMPTY = []
_FLAT_OWNERS = weakref.WeakValueDictionary()


def _adam_group_is_plain(g):
    return not (g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or g.get("foreach")) \
        and isinstance(g["lr"], float) and g.get("decoupled_weight_decay") in (None, False)


def _adam_adopt(opt):
    """-> _AdoptedAdam when `opt` is a stock Adam with ONE group holding exactly one qpnet_amd module's parameters in order (what
    `torch.optim.Adam(model.parameters(), ...)` builds, reference src/bin/qpnet_train.py:426-429), else None.  The optimizer's state becomes VIEWS of two
    flat moment buffers (a state loaded from a checkpoint is copied into them first), so state_dict() / load_state_dict() keep torch's layout and a
    reference-made checkpoint resumes here and vice versa."""
    if len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    ps = g["params"]
    if not ps or not _adam_group_is_plain(g):
        return None
    model = _FLAT_OWNERS.get(id(ps[0])) if ps[0].__dict__.get("_qpn_flat_view") else None
    if model is None:
        return None
    params = model_params(model)
    flat = getattr(model, "_flat", None)
    if flat is None or not flat.is_cuda or len(params) != len(ps) or any(a is not b for a, b in zip(params, ps)):
        return None
    a = _AdoptedAdam()
    a.model, a.n, a.stash = weakref.ref(model), len(ps), None
    a.m, a.v = torch.zeros_like(flat), torch.zeros_like(flat)
    a.mviews, a.vviews = _split_like(model, a.m), _split_like(model, a.v)
    steps = None
    for p, mv, vv in zip(ps, a.mviews, a.vviews):
        st = opt.state.get(p)
        if not st:
            k = 0
        else:
            if set(st) != {"step", "exp_avg", "exp_avg_sq"}:
                return None
            k = int(float(st["step"]))                    # (a read-back per parameter: once per adoption, i.e. once per optimizer or load_state_dict)
            mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])
        if steps is not None and k != steps:
            return None                                   # parameters stepped a different number of times (some had no gradient): torch's own loop keeps them
        steps = k
    a.steps = steps
    a.step_t = torch.tensor(float(steps), dtype=torch.float32)
    if steps > 0:
        _adam_bind_state(opt, a, ps)
    if not opt.__dict__.get("_qpn_sd_hook"):
        opt.__dict__["_qpn_sd_hook"] = True
        opt.register_state_dict_post_hook(_adam_state_dict_posthook)
    return a


def _adam_state_dict_posthook(opt, sd):
    """state_dict() of an adopted optimizer: every parameter gets a step counter of its OWN (inside the optimizer they all are one host scalar, bumped once
    per step; whoever loads the checkpoint -- torch's own Adam in the reference trainer -- increments them one by one, in place)."""
    a = opt.__dict__.get("_qpn_adopt")
    if a and a.steps > 0:
        sd["state"] = {k: (dict(st, step=torch.tensor(float(a.steps), dtype=torch.float32)) if st.get("step") is a.step_t else st) for k, st in sd["state"].items()}
    return sd


def _adam_bind_state(opt, a, ps):
    for p, mv, vv in zip(ps, a.mviews, a.vviews):
        opt.state[p] = {"step": a.step_t, "exp_avg": mv, "exp_avg_sq": vv}


def _adam_release(opt, a, ps):
    """hand the state back to torch's own implementation (fused=True wants a step counter per parameter, on the device).  Only state the hooks themselves
    own is rewritten -- entries whose step counter IS the adopter's shared host scalar.  A state that `load_state_dict()` has put in their place (a rollback to
    an earlier checkpoint, or an empty state) is left exactly as loaded: its own step counts drive the bias correction from here on, and the next step adopts
    it afresh (the adopter's old count used to overwrite the loaded one, and an empty loaded state got a lone 'step' key)."""
    if a.steps > 0:
        for p in ps:
            st = opt.state.get(p)                         # (.get: the state is a defaultdict -- indexing would create an entry)
            if st and st.get("step") is a.step_t:
                st["step"] = torch.tensor(float(a.steps), dtype=torch.float32, device=p.device)
    opt.__dict__["_qpn_adopt"] = None                     # looked at again at the next step


def _adam_prehook(opt, args, kwargs):
    """The reference trainer builds `torch.optim.Adam(model.parameters(), lr=...)` itself (src/bin/qpnet_train.py:426-429) and steps it once per chunk
    (:531).  With torch's default (foreach) implementation that is ~0.7 ms of host time per step for this model's 120 small tensors, with `fused=True`
    still ~0.25 ms (the per-parameter state walk in Python) -- most of what the unchanged loop spends (tools/dropin_prof.py).  This global step pre-hook
    makes the caller's stock Adam step through the library's ONE Adam kernel over the flat parameter buffer (qpn_adam_step: same update rule, tested
    against torch's, tests/test_train_gpu.py) when that is exactly equivalent:
      * type(opt) is torch.optim.Adam, one group, its parameters = one qpnet_amd module's parameters in order, no amsgrad / maximize / capturable /
        differentiable / foreach / tensor lr, no closure;
      * every p.grad is the view the module's backward handed out (_anchored_backward) -- clipped or rescaled in place is fine; a gradient that was
        replaced is gathered; a missing one (torch skips that parameter) leaves the step to torch.
    The optimizer's state is kept as views of two flat moment buffers (_adam_adopt), the group's hyper-parameters are read every step (lr schedulers
    work unchanged), and while torch's own `step` body runs the group's parameter list is empty, so it does nothing (restored by the post-hook).
    Everything else -- other optimizers, subclasses, several groups -- falls back to `fused=True` on eligible groups or is left alone.
    QPN_DROPIN_FUSED_ADAM=0 switches the hooks off, =torch keeps only the `fused=True` switch."""
    if type(opt) is not torch.optim.Adam:
        return
    d = opt.__dict__
    a = d.get("_qpn_adopt")
    if a is False:
        return
    has_closure = (args[1] if len(args) > 1 else kwargs.get("closure")) is not None       # (args = (optimizer, *step's positional arguments))
    g0 = opt.param_groups[0] if opt.param_groups else None
    if a is not None and a.stash is not None:            # a step that raised between the hooks
        g0["params"] = a.stash; a.stash = None
    if a is None:
        mode = os.environ.get("QPN_DROPIN_FUSED_ADAM", "1")
        if mode == "0":
            d["_qpn_adopt"] = False
            return
        if not d.get("_qpn_checked"):
            d["_qpn_checked"] = True
            _adam_switch_to_fused(opt)
        if mode == "torch" or has_closure:
            return
        try:
            a = _adam_adopt(opt)
        except Exception:                                # (an optimizer state this code does not know: torch's own step, as if the hooks were off)
            a = None
        if a is None:
            d["_qpn_adopt"] = False
            return
        d["_qpn_adopt"] = a
    ps = g0["params"]
    model = a.model()
    if has_closure or len(opt.param_groups) != 1 or model is None or len(ps) != a.n or not _adam_group_is_plain(g0):
        return _adam_release(opt, a, ps)
    flat = getattr(model, "_flat", None)
    if flat is None or flat.numel() != a.m.numel() or flat.device != a.m.device or ps[0].data_ptr() != flat.data_ptr():
        return _adam_release(opt, a, ps)
    c = model.__dict__.get("_qpn_gflat")
    gbuf = None
    if c is not None:
        for p, v in zip(ps, c[2]):
            if p.grad is not v:
                break
        else:
            gbuf = c[1]
    if gbuf is None:
        if any(p.grad is None for p in ps):
            return _adam_release(opt, a, ps)
        gbuf = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in ps])
    if a.steps > 0:
        st = opt.state.get(ps[0]); st1 = opt.state.get(ps[-1])
        if not st or not st1 or st.get("exp_avg") is not a.mviews[0] or st1.get("exp_avg_sq") is not a.vviews[-1] or st["step"] is not a.step_t:
            return _adam_release(opt, a, ps)              # load_state_dict() replaced the state: adopted again at the next step
    dev = flat.device
    L, hd = model._native(dev)
    b1, b2 = g0["betas"]
    with torch.cuda.device(dev):
        _lib.check(L.qpn_train_status_collect(hd))       # a pending forward check (no-op after a backward has collected it)
        _lib.check(L.qpn_adam_step(hd, flat.data_ptr(), gbuf.data_ptr(), a.m.data_ptr(), a.v.data_ptr(), flat.numel(), a.steps + 1,
                                   g0["lr"], b1, b2, g0["eps"], g0["weight_decay"], torch.cuda.current_stream(dev).cuda_stream))
    a.steps += 1
    a.step_t += 1
    if a.steps == 1:
        _adam_bind_state(opt, a, ps)
    a.stash = ps
    g0["params"] = _EMPTY


def _adam_posthook(opt, args, kwargs):
    a = opt.__dict__.get("_qpn_adopt")
    if a and a.stash is not None:
        opt.param_groups[0]["params"] = a.stash
        a.stash = None


def _adam_switch_to_fused(opt):
    """eligible groups of a stock Adam -> torch's own fused implementation (what runs whenever _adam_prehook leaves a step to torch)."""
    for g in opt.param_groups:
        ps = g["params"]
        if not ps or g.get("fused") is not None or g.get("foreach") is not None or g.get("capturable") or g.get("differentiable") or g.get("amsgrad"):
            continue
        if not all(p.__dict__.get("_qpn_flat_view") and p.is_cuda and p.dtype == torch.float32 for p in ps):
            continue
        g["fused"] = True
        for p in ps:                                  # a state loaded from a checkpoint keeps its step counters on the host: the fused kernel wants them next to the parameter
            st = opt.state.get(p)
            if st and "step" in st and torch.is_tensor(st["step"]) and st["step"].device != p.device:
                st["step"] = st["step"].to(device=p.device, dtype=torch.float32)


from torch.optim.optimizer import register_optimizer_step_pre_hook as _register_optimizer_step_pre_hook  # noqa: E402
from torch.optim.optimizer import register_optimizer_step_post_hook as _register_optimizer_step_post_hook  # noqa: E402
_register_optimizer_step_pre_hook(_adam_prehook)
_register_optimizer_step_post_hook(_adam_posthook)


def join_staged(x, dev, *more):
    """Inputs copied to the device on a stream of their own (runners.PinnedStager: the copy must not sit between the step's kernels on the compute stream) carry
    {ready event, device buffer}: the consumer's stream waits for the copy -- long done when a prefetch thread is ahead, so the wait is a queue packet, not a
    stall -- and the buffer is marked as used on that stream (the caching allocator will not hand it out again before the step that reads it has run).
    Every consumer of loader batches calls this on ALL its inputs (FusedTrainer.step / forward_loss, QPNet.forward): a tensor that never
    went through a stager costs one dictionary look-up."""
    done = None
    for t in (x,) + more:
        st = t.__dict__.get("_qpn_staged") if isinstance(t, torch.Tensor) else None
        if st is None:
            continue
        if st[0] is not done:                          # (the tensors of one batch share one copy: one wait, one record_stream)
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(st[0])
            st[1].record_stream(cur)
            done = st[0]
        del t.__dict__["_qpn_staged"]                  # (once per batch)


def model_params(model):
    """list(model.parameters()), cached (see above)."""
    c = model.__dict__.get("_qpn_params")
    # (deletions -- `del model.x` -- fire no hook: the count of registered sub-modules / parameters of the top level is part of the key)
    key = (_PARAM_GENERATION[0], len(model._modules), len(model._parameters))
    if c is None or c[0] != key:
        c = (key, list(model.parameters()))
        model.__dict__["_qpn_params"] = c
    return c[1]


def ensure_flat(model, dev):
    """One contiguous fp32 buffer holding every parameter in state_dict order, with the
    nn.Parameters re-pointed at views of it (so optimizers / load_state_dict write into it and
    the C ABI reads it without a gather).  Rebuilt if .to()/.cuda() replaced the storages."""
    params = model_params(model)
    flat = getattr(model, "_flat", None)
    ok = flat is not None and flat.device == dev
    if ok:
        # 120 data_ptr() calls cost ~60 us of interpreter time per step: the full walk runs when the parameter list was rebuilt and every 32nd call; every
        # call looks at the first and the last view (.to() / .cuda() / load into new storage replace ALL of them) and at FOUR more in rotation, with their
        # dtype -- a single interior `p.data = ...` (pruning, re-parametrisation, a partial load with assign=True) is seen within 30 calls at the latest
        c = model.__dict__.get("_qpn_flat_chk")
        base = flat.data_ptr()
        if c is not None and c[0] is params and c[1] < 32:
            c[1] += 1
            last = params[-1]
            ok = params[0].data_ptr() == base and last.data_ptr() == base + 4 * (flat.numel() - last.numel()) and last.dtype == torch.float32
            offs, k, n = c[2], c[3], len(params)
            for _ in range(4):
                k = k + 1 if k + 1 < n else 0
                q = params[k]
                if q.data_ptr() != base + 4 * offs[k] or q.dtype != torch.float32:
                    ok = False
            c[3] = k
        else:
            o, offs = 0, []
            for p in params:
                if p.data_ptr() != base + 4 * o or p.dtype != torch.float32:
                    ok = False
                    break
                offs.append(o)
                o += p.numel()
            model.__dict__["_qpn_flat_chk"] = [params, 0, offs, 0] if ok else None
    if not ok:
        flat = torch.cat([p.detach().reshape(-1).to(dev, torch.float32) for p in params]).contiguous()
        o = 0
        _FLAT_OWNERS[id(params[0])] = model             # (looked up by _adam_adopt, which checks every parameter's identity against this module's)
        for p in params:
            n = p.numel()
            p.data = flat[o:o + n].view(p.shape)
            p.__dict__["_qpn_flat_view"] = True      # (what _adam_prehook recognises this module's parameters by; a plain bool: parameters get pickled)
            o += n
        model._flat = flat
    return model._flat


def _split_like(model, gflat):
    """views of the flat gradient, one per parameter: ONE split call + a reshape for the tensors that are not 1-D (slicing and viewing
    each of the 120 by hand was 0.27 ms of the backward's host time)."""
    params = model_params(model)
    c = model.__dict__.get("_qpn_split")
    if c is None or c[0] is not params:
        c = (params, [p.numel() for p in params], [tuple(p.shape) if p.dim() != 1 else None for p in params])
        model.__dict__["_qpn_split"] = c
    pieces = gflat.split_with_sizes(c[1])
    return [g if shp is None else g.view(shp) for g, shp in zip(pieces, c[2])]


def _flat_grad_of(params):
    """The flat gradient buffer the parameters' .grad tensors are consecutive views of, or None."""
    if not params or params[0].grad is None:
        return None
    base = params[0].grad.data_ptr()
    o = 0
    for p in params:
        g = p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + 4 * o:
            return None
        o += p.numel()
    g0 = params[0].grad
    return torch.as_strided(g0, (o,), (1,), g0.storage_offset()) if g0.untyped_storage().nbytes() >= 4 * (g0.storage_offset() + o) else None


class QPNetFunction(torch.autograd.Function):
    """logits = QPNet.forward(x, h, d, blength); backward = hand-written HIP kernels."""

    @staticmethod
    def forward(ctx, model, x, h, d, BL, maxd, *params):
        dev = x.device
        L, hd = model._native(dev)
        flat = model._flat                                    # (qpnet_forward has just validated it)
        B, T = x.shape
        logits = torch.empty((B, BL, model.n_quantize), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            # The reference asserts on the gather bounds inside forward (qpnet.py:294).  Here the device-side check is enqueued behind
            # the kernels and collected (no stream drain) at the first of: this forward's backward (so an out-of-range factor is raised
            # BEFORE the optimizer step it would feed), FlatAdam.step, the next forward, model.check_status().  A forward that records no
            # graph (torch.no_grad() / eval loops: nothing else of this model may ever run) checks synchronously.
            _lib.check(L.qpn_train_status_collect(hd))
            _lib.check(L.qpn_train_forward(hd, flat.data_ptr(), B, T, h.shape[2], d.shape[1], BL, maxd,
                                           x.data_ptr(), h.data_ptr(), d.data_ptr(), logits.data_ptr(), stream))
            if model._qpn_sync_status:
                _lib.check(L.qpn_train_status(hd, stream))
            else:
                _lib.check(L.qpn_train_status_enqueue(hd, stream))
        ctx.model = model
        ctx.anchored = len(params) == 1 and params[0] is model.__dict__.get("_qpn_anchor")
        ctx.generation = int(L.qpn_train_generation(hd))
        ctx.keep = (x, h, d)                                  # inputs must outlive backward (the workspace points into them)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        dev = dlogits.device
        L, hd = model._native(dev)
        if int(L.qpn_train_generation(hd)) != ctx.generation:
            raise RuntimeError("qpnet_amd: backward of a forward whose activations have been replaced by a later forward of the "
                               "same model (one outstanding forward per model; run validation forwards after backward)")
        with torch.cuda.device(dev):
            # the forward's gather-bounds / target check.  Default: WAIT for its copy (it sits behind the forward's kernels, so a host that runs ahead of
            # the device blocks here until the forward has drained) -- the one way to raise BEFORE an optimizer this module does not control (stock
            # torch.optim.Adam) steps on a flagged gradient, as the reference's in-line assert does (qpnet.py:294).  model.status_check = "lazy": only
            # what has already landed is looked at (hipEventQuery); the rest is collected by FlatAdam.step / the next forward / model.check_status(),
            # and the fused Adam kernel skips flagged updates on the device by itself.
            if getattr(model, "status_check", "backward") == "lazy":
                _lib.check(L.qpn_train_status_poll(hd, None))
            else:
                _lib.check(L.qpn_train_status_collect(hd))
        flat = model._flat
        dl = dlogits.contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        if ctx.anchored:
            _anchored_backward(model, L, hd, dl, flat, stream, dev)
            return (None, None, None, None, None, None, None)
        # a FRESH buffer per backward: autograd keeps (or accumulates into) the views it is handed, so they must not
        # alias memory a later backward writes
        g = torch.empty_like(flat)
        with torch.cuda.device(dev):
            _lib.check(L.qpn_train_backward(hd, dl.data_ptr(), g.data_ptr(), stream))
        return (None, None, None, None, None, None) + tuple(_split_like(model, g))


def _anchored_backward(model, L, hd, dl, flat, stream, dev):
    """The reference loop's backward without autograd's 120-way fan-out (qpnet_train.py:527-531: zero_grad, loss.backward(), optimizer.step()).
    The graph holds ONE anchor leaf instead of the 120 parameters, and this function gives every parameter its .grad itself, with
    loss.backward()'s semantics: parameters without a gradient (after optimizer.zero_grad(), whose default sets them to None) get one, the
    others accumulate.  The gradients are consecutive views of one persistent flat buffer the kernels write directly (NOTE the aliasing: a reference
    to a p.grad kept across zero_grad(set_to_none=True) is overwritten by the next backward, unlike torch's fresh tensors), so the usual step
    costs one launch sequence and 120 attribute stores -- measured on the host: 0.63 -> 0.25 ms of the 1.63 ms the unchanged loop spends per step
    (autograd built 120 views and ran 120 AccumulateGrad nodes).  Not visible to this path: torch.autograd.grad() w.r.t. parameters (torch
    reports them as unused in the graph) and tensor hooks on parameters; QPN_DROPIN_FLAT_GRAD=0 hands autograd the parameters as before."""
    params = model_params(model)
    c = model.__dict__.get("_qpn_gflat")
    if c is None or c[0] is not params or c[1].device != dev or c[1].numel() != flat.numel():
        g = torch.zeros_like(flat)
        c = (params, g, _split_like(model, g))
        model.__dict__["_qpn_gflat"] = c
    gflat, views = c[1], c[2]
    none_yet = params[0].grad is None and all(p.grad is None for p in params)
    mine = False
    if not none_yet:
        fb = _flat_grad_of(params)
        mine = fb is not None and fb.data_ptr() == gflat.data_ptr()
    with torch.cuda.device(dev):
        if none_yet:                                       # the usual step: the kernels write the buffer the .grad views live in
            _lib.check(L.qpn_train_backward(hd, dl.data_ptr(), gflat.data_ptr(), stream))
            for p, v in zip(params, views):
                p.grad = v
            return
        tmp = torch.empty_like(flat)
        _lib.check(L.qpn_train_backward(hd, dl.data_ptr(), tmp.data_ptr(), stream))
    if mine:                                               # accumulation (two backwards, or zero_grad(set_to_none=False)): one add
        gflat.add_(tmp)
        return
    for p, v, t in zip(params, views, _split_like(model, tmp)):      # somebody replaced some of the gradients: one by one
        if p.grad is None:
            v.copy_(t); p.grad = v
        else:
            p.grad.add_(t)


def forward_maxd(model, T, F, Td, BL, dilated_factors):
    """ceil(max d) as QPNet.forward needs it (reference qpnet.py:254-262: it sets the receptive field RF = RF_A * maxd +
    RF_F + 1 and the chunk's last RF + BL samples are used) WITHOUT reading the factors back from the device.

    The largest maxd the chunk's shapes admit is used instead: RF_A * maxd + RF_F + 1 + BL <= min(T, Td + 1, F * U + 1).  Any
    maxd >= the true ceil(max d) gives the same logits -- the extra leading rows are context the last BL outputs do not see,
    and the pitch taps are positions counted from the END of the chunk -- while a true maximum that does NOT fit the chunk
    is what the reference's gather assert (qpnet.py:294) catches; here the device-side bound check reports it.  The
    reference trainer cuts chunks of exactly RF + BL samples (qpnet_train.py:268-284), for which this IS ceil(max d).
    `model.read_back_maxd = True` restores the device read-back (one stream synchronisation per forward)."""
    recA, recF = model.receptiveA_field, model.receptiveF_field
    if getattr(model, "read_back_maxd", False) or recA <= 0:
        return int(torch.max(dilated_factors.ceil()))
    U = model.upsampling_factor
    avail = min(int(T), int(Td) + 1, (int(F) * U if U > 0 else int(F)) + 1)
    return max((avail - recF - 1 - int(BL)) // recA, 1)


def qpnet_forward(model, x, h, dilated_factors, blength):
    """QPNet.forward (reference qpnet.py:239-262 for the argument handling)."""
    dev = x.device
    if dev.type != "cuda":
        raise RuntimeError("qpnet_amd.QPNet runs on an AMD GPU only (tensors are on %s); there is no CPU fallback" % dev)
    # (one read-back when blength lives on the device, as in the reference trainer; a host tensor / array costs nothing)
    bl_host = blength.tolist() if hasattr(blength, "tolist") else list(blength)
    assert all(v == bl_host[0] for v in bl_host)
    BL = int(bl_host[0])
    join_staged(x, dev, h, dilated_factors)
    maxd = forward_maxd(model, x.shape[1], h.shape[2], dilated_factors.shape[1], BL, dilated_factors)
    x = x.to(dev, torch.int64).contiguous()
    h = h.to(dev, torch.float32).contiguous()
    d = dilated_factors.to(dev, torch.float32).contiguous()
    ensure_flat(model, dev)
    params = model_params(model)
    # no graph will be recorded (no_grad, or no parameter wants a gradient): no backward will come to collect the status
    wants = [p.requires_grad for p in params]
    model._qpn_sync_status = not (torch.is_grad_enabled() and any(wants))
    if torch.is_grad_enabled() and all(wants) and os.environ.get("QPN_DROPIN_FLAT_GRAD", "1") != "0":
        # every parameter trains (the reference trainer): ONE anchor leaf stands for them in the graph, the backward assigns their gradients (_anchored_backward)
        a = model.__dict__.get("_qpn_anchor")
        if a is None or a.device != dev:
            a = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)
            model.__dict__["_qpn_anchor"] = a
        return QPNetFunction.apply(model, x, h, d, BL, maxd, a)
    return QPNetFunction.apply(model, x, h, d, BL, maxd, *params)


# ---------------------------------------------------------------- Adam state in torch.optim.Adam's state_dict layout
_ADAM_GROUP_DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)


def adam_state_to_torch(model, m, v, steps, hyper):
    """{"state": {i: {"step","exp_avg","exp_avg_sq"}}, "param_groups": [...]}: what torch.optim.Adam(model.parameters())
    .state_dict() holds after `steps` steps, built from the flat moment buffers (reference checkpoints store exactly this,
    src/bin/qpnet_train.py:346-352), so either side can resume the other's checkpoint."""
    params = list(model.parameters())
    state, o = {}, 0
    for i, p in enumerate(params):
        n = p.numel()
        if steps > 0 and m is not None:
            state[i] = {"step": torch.tensor(float(steps)), "exp_avg": m[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": v[o:o + n].view(p.shape).clone()}
        o += n
    group = dict(lr=hyper["lr"], betas=tuple(hyper["betas"]), eps=hyper["eps"], weight_decay=hyper["weight_decay"])
    group.update(_ADAM_GROUP_DEFAULTS)
    group["params"] = list(range(len(params)))
    return {"state": state, "param_groups": [group]}


def adam_state_from_torch(model, sd, flat):
    """inverse of adam_state_to_torch; also accepts the previous round's {"flat_adam": ...} format."""
    if "flat_adam" in sd:
        st = sd["flat_adam"]
        return st["exp_avg"].to(flat.device).clone(), st["exp_avg_sq"].to(flat.device).clone(), int(st["step"]), dict(sd["param_groups"][0])
    params = list(model.parameters())
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    steps, o = 0, 0
    index = sd["param_groups"][0].get("params", list(range(len(params))))
    for pos, p in enumerate(params):
        n = p.numel()
        st = sd["state"].get(index[pos]) if pos < len(index) else None
        if st is not None:
            m[o:o + n] = st["exp_avg"].reshape(-1).to(flat.device, torch.float32)
            v[o:o + n] = st["exp_avg_sq"].reshape(-1).to(flat.device, torch.float32)
            steps = max(steps, int(float(st["step"])))
        o += n
    hyper = {k: sd["param_groups"][0][k] for k in ("lr", "betas", "eps", "weight_decay") if k in sd["param_groups"][0]}
    if sd["param_groups"][0].get("amsgrad"):
        raise ValueError("amsgrad Adam state cannot be resumed by the fused Adam kernel")
    return m, v, steps, hyper


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (lr, betas, eps, weight_decay; no amsgrad) as ONE kernel over the model's flat parameter
    buffer -- a one-line swap for `torch.optim.Adam(model.parameters(), lr=...)` in the reference loop
    (src/bin/qpnet_train.py:426-429,531), whose 120-tensor foreach update costs 0.9 ms per step on this GPU.
    Gradients are read from `p.grad`: when they are consecutive views of one flat buffer (what the autograd backward hands
    out) no gather happens.  state_dict()/load_state_dict() use torch.optim.Adam's own layout, so checkpoints written by
    the reference trainer resume here and vice versa."""

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
        with torch.cuda.device(dev):
            _lib.check(L.qpn_train_status_collect(hd))       # a pending forward check (no-op after a backward has collected it)
        params = model_params(model)
        g = _flat_grad_of(params)
        if g is None:                                # grads came from elsewhere (or some are None): gather them
            g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
        if self._m is None or self._m.numel() != flat.numel() or self._m.device != dev:
            m0, v0 = self._m, self._v
            self._m = torch.zeros_like(flat); self._v = torch.zeros_like(flat)
            if m0 is not None and m0.numel() == flat.numel():
                self._m.copy_(m0); self._v.copy_(v0)
        grp = self.param_groups[0]
        self._steps += 1
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(L.qpn_adam_step(hd, flat.data_ptr(), g.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), flat.numel(),
                                       self._steps, grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"], stream))
        return loss

    def _hyper(self):
        return {k: self.param_groups[0][k] for k in ("lr", "betas", "eps", "weight_decay")}

    def state_dict(self):
        return adam_state_to_torch(self.model, self._m, self._v, self._steps, self._hyper())

    def load_state_dict(self, sd):
        params = list(self.model.parameters())
        flat = getattr(self.model, "_flat", None)
        if flat is None:
            flat = torch.cat([p.detach().reshape(-1).float() for p in params])
        self._m, self._v, self._steps, hyper = adam_state_from_torch(self.model, sd, flat)
        for k, v in hyper.items():
            if k != "params":
                self.param_groups[0][k] = v


class FusedTrainer:
    """forward + mean CE + backward + Adam entirely behind the C ABI (flat fp32 buffers).

    Mirrors the loop body of the reference trainer (src/bin/qpnet_train.py:517-531) with
    torch.optim.Adam(lr=1e-4, betas=(0.9,0.999), eps=1e-8, weight_decay=0) semantics.
    With world_size > 1 every rank's gradient is weighted by its row count inside the gradient-reduction kernel, summed
    over ranks with ONE RCCL all-reduce of the flat buffer (the row counts ride in its last element) and divided by the
    global row count inside the Adam kernel: the update is the gradient of the mean CE over ALL ranks' rows, with no
    extra elementwise launch or allocation in the step.
    state_dict()/load_state_dict() use torch.optim.Adam's layout (resume of reference-made checkpoints and vice versa)."""

    # The device-side status word (bad taps / targets; the reference asserts in-line, qpnet.py:294, qpnet_train.py:525) is copied to pinned
    # memory behind every step; a step starts by looking at the copy made TWO steps earlier (the previous step is still queued on the
    # device while the host enqueues this one: waiting for it would idle the GPU).  A bad chunk is raised two steps late at most, not up to 99
    # as when the word was read every 100 steps.  check_status() collects everything outstanding.

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None, world_size=1):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.step_count = 0
        self.m = self.v = self.g = None
        self.pg, self.world = process_group, world_size
        self._logits = None
        self.last_buckets = (0, 0)
        self._applied_base = None     # (step_count, the handle's applied-update count) at this trainer's first step / after a re-base (see _rebase_step_count)
        self._early = None            # the stream the early bucket of the gradient exchange runs on (world_size > 1)
        self._two_buckets = None      # agreed over the process group at the first data-parallel step (every rank must issue the same collectives)

    def _buffers(self, flat):
        if self.m is None or self.m.device != flat.device or self.m.numel() != flat.numel():
            m0, v0 = self.m, self.v
            self.m = torch.zeros_like(flat); self.v = torch.zeros_like(flat)
            if m0 is not None and m0.numel() == flat.numel():
                self.m.copy_(m0); self.v.copy_(v0)
        if self.g is None or self.g.device != flat.device or self.g.numel() != flat.numel() + 4:
            self.g = torch.empty(flat.numel() + 4, dtype=torch.float32, device=flat.device)   # gradient + trailer (row count)

    @staticmethod
    def _norm(t, dtype, dev):
        return t if (t.dtype == dtype and t.device == dev and t.is_contiguous()) else t.to(dev, dtype).contiguous()

    def step(self, x, h, t, d, blength, want_loss=True, maxd=None):
        """One optimisation step.  maxd = ceil(max(d)) of the chunk; a loader that built d on the host (the reference's
        train_generator does, qpnet_train.py:268-272) passes it in, otherwise it is read back from the device (one sync).
        Inputs are normalised like QPNet.forward does (int64 samples/targets, float32 features/factors, contiguous)."""
        model = self.model
        dev = x.device
        if dev.type != "cuda":
            raise RuntimeError("qpnet_amd.FusedTrainer runs on an AMD GPU only (tensors are on %s)" % dev)
        L, hd = model._native(dev)
        flat = ensure_flat(model, dev)
        self._buffers(flat)
        join_staged(x, dev, h, t, d)
        x = self._norm(x, torch.int64, dev); t = self._norm(t, torch.int64, dev)
        h = self._norm(h, torch.float32, dev); d = self._norm(d, torch.float32, dev)
        assert x.dim() == 2 and t.shape[0] == x.shape[0] and h.dim() == 3 and h.shape[1] == model.n_aux and d.dim() == 2
        BL = int(blength[0])
        if maxd is None:
            maxd = int(torch.max(d.ceil()))
        B, T = x.shape
        if self._logits is None or self._logits.shape != (B, BL, model.n_quantize) or self._logits.device != dev:
            self._logits = torch.empty((B, BL, model.n_quantize), dtype=torch.float32, device=dev)
            self._dlogits = torch.empty_like(self._logits)
        stream = torch.cuda.current_stream(dev).cuda_stream
        loss = C.c_double(0.0)
        multi = self.world > 1
        if self._applied_base is None and not torch.cuda.is_current_stream_capturing():
            n = C.c_int64(0)
            with torch.cuda.device(dev):
                _lib.check(L.qpn_train_applied_updates(hd, C.byref(n), stream))      # (once per trainer: drains the stream before its first step)
            self._applied_base = (self.step_count, int(n.value))
        # (a step being captured into a hipGraph -- torch.cuda.graph -- may not wait for events or read anything back: no status bookkeeping, the
        #  caller checks with check_status() outside the graph; the library runs the stack as a launch per layer while a stream is capturing)
        capturing = torch.cuda.is_current_stream_capturing()
        if not multi and not capturing:
            # the whole step behind ONE foreign call (qpn_train_step: the same calls in the same order as below): while it runs, the interpreter is free
            # for the loader thread -- the runner loop was bound by the two threads' interleaved Python, not by the device
            self.step_count += 1
            valid = C.c_int(0)
            mode = 1 if want_loss == "lagged" else (2 if want_loss else 0)
            with torch.cuda.device(dev):
                rc = L.qpn_train_step(hd, flat.data_ptr(), B, T, h.shape[2], d.shape[1], BL, maxd, x.data_ptr(), h.data_ptr(), d.data_ptr(),
                                      t.data_ptr(), t.shape[1], self._logits.data_ptr(), self._dlogits.data_ptr(), self.g.data_ptr(),
                                      self.m.data_ptr(), self.v.data_ptr(), flat.numel(), self.step_count, self.lr, self.betas[0], self.betas[1],
                                      self.eps, self.wd, mode, C.byref(loss), C.byref(valid), stream)
            if rc:
                self._rebase_step_count(L, hd, dev, stream, rc)
            return loss.value if valid.value else None
        try:
            return self._step_calls(L, hd, dev, flat, stream, x, h, t, d, B, T, BL, maxd, want_loss, multi, capturing, loss)
        except _lib.QpnError as e:
            if e.code in (-4, -2):                     # (QPN_ERANGE / QPN_ENODEV: what the device-side status word raises)
                self._rebase_step_count(L, hd, dev, stream, 0)
            raise

    def _rebase_step_count(self, L, hd, dev, stream, rc):
        """A device-side status error: the Adam kernel skipped the flagged step's update and that of every step enqueued behind it until the host collected the
        word (k_adam), so the host's step count -- the bias correction's exponent, and what checkpoints store -- has run ahead of the updates applied.  It is set back
        to (count at the base) + (updates the device applied since): one stream drain, on the error path only.  rc != 0: raise it afterwards."""
        msg = L.qpn_last_error()                       # (the call below would replace it)
        if self._applied_base is not None:
            n = C.c_int64(0)
            with torch.cuda.device(dev):
                if L.qpn_train_applied_updates(hd, C.byref(n), stream) == 0:
                    self.step_count = self._applied_base[0] + int(n.value) - self._applied_base[1]
        if rc:
            raise _lib.QpnError(rc, msg.decode("utf-8", "replace"))

    def _step_calls(self, L, hd, dev, flat, stream, x, h, t, d, B, T, BL, maxd, want_loss, multi, capturing, loss):
        with torch.cuda.device(dev):
            if not capturing:
                _lib.check(L.qpn_train_status_collect_lagged(hd))   # the check of the step before the previous one (never waits for queued work)
            # forward + CrossEntropyLoss + dL/dlogits in one call (the loss stays on the device unless asked for)
            _lib.check(L.qpn_train_forward_loss(hd, flat.data_ptr(), B, T, h.shape[2], d.shape[1], BL, maxd,
                                                x.data_ptr(), h.data_ptr(), d.data_ptr(), t.data_ptr(), t.shape[1],
                                                self._logits.data_ptr(), 2, self._dlogits.data_ptr(), stream))      # (2: the backward of this forward follows, same dL/dlogits)
            if multi:
                from .parallel import exchange_two_buckets
                first, count = C.c_int64(0), C.c_int64(0)
                if self._early is None:
                    self._early = torch.cuda.Stream(dev)
                # (opt-in, QPN_EXCHANGE_BUCKETS=2.  Measured with RCCL at one rank: the second call costs 18 us of stream time, 0.841 -> 0.859 ms a
                #  step; what it can hide at 8 ranks is a quarter of the bandwidth term of a 2.3 MB, latency-bound all-reduce -- DESIGN.md section 7)
                if self._two_buckets is None:
                    # How many collectives a step issues must not be a per-rank decision (environment and launch-plan knobs can differ per rank: a
                    # rank that sends one all-reduce while its peers send two deadlocks RCCL or corrupts the sum): the split is agreed ONCE, as the
                    # minimum over the ranks of "this rank wants and can do two buckets"; thereafter every rank issues two all-reduces over the SAME
                    # ranges in every step, whether or not its early range happened to be ready early in that step.
                    self._two_buckets = self._agree_two_buckets(L, hd, flat)
                # g <- n_r * grad_r with {n_r, flagged_r} appended; SUM over ranks; Adam divides by the summed row count on the device and skips the update on
                # EVERY rank when any rank's status word was set (2: the trailer leaves with the early bucket, before the backward's last launch could rewrite it)
                _lib.check(L.qpn_train_backward_ex(hd, self._dlogits.data_ptr(), self.g.data_ptr(), float(B * BL), 2 if self._two_buckets else 1, stream))
                # two buckets: the tail the side stream has finished already can go out under the layer backward (qpn_train_early_bucket)
                if self._two_buckets:
                    _lib.check(L.qpn_train_early_bucket(hd, C.byref(first), C.byref(count), self._early.cuda_stream))
                    if count.value == 0:      # nothing was finished early in this step (one stream: a profile is being taken): same two ranges, after the backward
                        first.value, count.value = self._two_buckets[0], self._two_buckets[1]
                        self._early.wait_stream(torch.cuda.current_stream(dev))
                exchange_two_buckets(self.g, first.value, count.value, self.pg, self._early if count.value else None)
                self.last_buckets = (first.value, count.value)       # (what the last step exchanged early; (0, 0): one exchange)
                L.qpn_train_profile_mark(hd, 9, stream)      # QPN_PG_ALLREDUCE (no-op unless a profile is being taken)
            else:
                _lib.check(L.qpn_train_backward(hd, self._dlogits.data_ptr(), self.g.data_ptr(), stream))
            self.step_count += 1
            _lib.check(L.qpn_adam_step_ex(hd, flat.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), flat.numel(),
                                          self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                          self.g.data_ptr() + 4 * flat.numel() if multi else None, stream))
            if want_loss == "lagged":
                # this step's loss is copied out behind its kernels; what comes back is the PREVIOUS step's (None at the first step, or right after
                # flush_loss()): the stream is never drained, and a caller that sums losses over an interval and calls flush_loss() at its end has the same sum
                _lib.check(L.qpn_train_loss_enqueue(hd, stream))
                _lib.check(L.qpn_train_status_enqueue(hd, stream))
                valid = C.c_int(0)
                _lib.check(L.qpn_train_loss_collect(hd, 0, C.byref(loss), C.byref(valid)))
                return loss.value if valid.value else None
            if want_loss:
                _lib.check(L.qpn_train_loss(hd, C.byref(loss), stream))
                _lib.check(L.qpn_train_status(hd, stream))   # (the stream has just been drained for the loss: in-step, like the reference)
            elif not capturing:
                _lib.check(L.qpn_train_status_enqueue(hd, stream))
        return loss.value if want_loss else None

    def _agree_two_buckets(self, L, hd, flat):
        """(first, count) of the early bucket if EVERY rank asked for two buckets (QPN_EXCHANGE_BUCKETS=2) and reports the same range, else False."""
        import torch.distributed as dist
        want = os.environ.get("QPN_EXCHANGE_BUCKETS", "1") == "2"
        first = int(L.qpn_train_early_first(hd)) if want else -1
        ok = want and first > 0
        if dist.is_initialized() and dist.get_world_size(self.pg) > 1:
            v = torch.tensor([first if ok else -1, -(first if ok else -1)], dtype=torch.int64, device=flat.device if dist.get_backend(self.pg) == "nccl" else "cpu")
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=self.pg)      # min(first) == max(first) > 0  <=>  every rank reports the same range
            ok = int(v[0]) > 0 and int(v[0]) == -int(v[1])
        return (first, flat.numel() - first + 4) if ok else False

    def flush_loss(self):
        """The loss of the last step(want_loss="lagged") (waits for that step), or None if it has been returned already."""
        L, hd = self.model._native(self.model._flat.device)
        loss, valid = C.c_double(0.0), C.c_int(0)
        _lib.check(L.qpn_train_loss_collect(hd, 1, C.byref(loss), C.byref(valid)))
        return loss.value if valid.value else None

    def check_status(self):
        """Raise what the device-side check of the last step(want_loss=False) found (see QPNet.check_status); the step count is set back to the
        updates the device actually applied (_rebase_step_count)."""
        try:
            self.model.check_status()
        except _lib.QpnError as e:
            flat = getattr(self.model, "_flat", None)
            if e.code in (-4, -2) and flat is not None and flat.is_cuda:
                L, hd = self.model._native(flat.device)
                self._rebase_step_count(L, hd, flat.device, torch.cuda.current_stream(flat.device).cuda_stream, 0)
            raise

    def forward_loss(self, x, h, t, d, blength, maxd=None):
        """forward + mean CE only (validation, reference qpnet_validate.py:409-430)."""
        model = self.model
        dev = x.device
        L, hd = model._native(dev)
        flat = ensure_flat(model, dev)
        join_staged(x, dev, h, t, d)                 # (run_validate feeds staged batches: the copy stream must be joined before anything reads them)
        x = self._norm(x, torch.int64, dev); t = self._norm(t, torch.int64, dev)
        h = self._norm(h, torch.float32, dev); d = self._norm(d, torch.float32, dev)
        BL = int(blength[0])
        if maxd is None:
            maxd = int(torch.max(d.ceil()))
        B, T = x.shape
        logits = torch.empty((B, BL, model.n_quantize), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        loss = C.c_double(0.0)
        with torch.cuda.device(dev):
            _lib.check(L.qpn_train_forward_loss(hd, flat.data_ptr(), B, T, h.shape[2], d.shape[1], BL, maxd,
                                                x.data_ptr(), h.data_ptr(), d.data_ptr(), t.data_ptr(), t.shape[1],
                                                logits.data_ptr(), 0, None, stream))
            _lib.check(L.qpn_train_loss(hd, C.byref(loss), stream))
            _lib.check(L.qpn_train_status(hd, stream))
        return loss.value

    # ---- optimizer-state interface (checkpoints: loaders.save_checkpoint(dir, model, trainer, it))
    def _hyper(self):
        return dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.wd)

    def state_dict(self):
        return adam_state_to_torch(self.model, self.m, self.v, self.step_count, self._hyper())

    def load_state_dict(self, sd):
        params = list(self.model.parameters())
        flat = getattr(self.model, "_flat", None)
        if flat is None or flat.device != params[0].device:
            flat = torch.cat([p.detach().reshape(-1).float() for p in params])
        self.m, self.v, self.step_count, hyper = adam_state_from_torch(self.model, sd, flat)
        self.lr = hyper.get("lr", self.lr); self.betas = tuple(hyper.get("betas", self.betas))
        self.eps = hyper.get("eps", self.eps); self.wd = hyper.get("weight_decay", self.wd)
