"""Helpers shared by the parity tests (inputs regenerated from seeds via qpnet_amd.synth)."""
import numpy as np
from qpnet_amd import synth, harness


decode_batch = synth.decode_batch


def build_model(cfg, flat, device):
    import torch
    from qpnet_amd.qpnet import QPNet
    m = QPNet(**cfg.kwargs())
    sd = {k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()}
    m.load_state_dict(sd)
    return m.to(device).eval()


def run_giveup_child(which, env_name):
    """The fault-injection hooks exist only in the -DQPN_TESTING build of the library (qpnet_amd/libqpnet_hip_testing.so, built by
    __graft_entry__.build()): the scenario runs in a child process bound to that build (tests/giveup_child.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "qpnet_amd", "libqpnet_hip_testing.so")
    assert os.path.exists(lib), "build the testing library first: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, QPN_LIB=lib, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env[env_name] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "giveup_child.py"), which], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GIVEUP_CHILD_OK " + which in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-5, a_rel=1e-4, kink_eps=4e-6, max_units=6):
    """Per-tensor comparison of a flat gradient with the numpy oracle's hand-derived backward: |g - ref| <= a_scale * max|ref (all tensors)| +
    a_rel * max|ref (this tensor)|.

    ReLU kinks: where a post-net pre-activation of the oracle's forward lies within fp32 reassociation error of zero (|s0| or |y0| < kink_eps;
    the logits of two correct fp32 implementations differ by ~1e-6), which side of the kink a unit falls on is decided by summation order, not
    by the algorithm -- and the side changes that row's gradient by a whole, small, term.  For those units (at most `max_units`) either side is
    accepted: the gradient must match the oracle's backward for ONE assignment of sides.  Returns the oracle gradient that matched."""
    import itertools
    import numpy as np
    offs, _ = cfg.param_offsets()

    def worst(og):
        scale = np.abs(og).max()
        bad = None
        for k, (o, shp) in offs.items():
            n = int(np.prod(shp))
            a, r = grad[o:o + n], og[o:o + n]
            e, bound = np.abs(a - r).max(), a_scale * scale + a_rel * np.abs(r).max()
            if e > bound and (bad is None or e / bound > bad[1]):
                bad = ("%s: err %.3e > bound %.3e" % (k, e, bound), e / bound)
        return bad

    og = TO.backward(cfg, flat, caches, dl)
    bad = worst(og)
    if bad is None:
        return og
    units = [(b, key, idx) for b, c in enumerate(caches) for key in ("s0", "y0") for idx in zip(*np.nonzero(np.abs(c[key]) < kink_eps))]
    assert units, "grad mismatch in %s (no pre-activation within %.0e of a ReLU kink)" % (bad[0], kink_eps)
    saved = [caches[b][key][idx] for b, key, idx in units]
    if len(units) > max_units:
        # too many near-kink units to enumerate their sides (a full-size chunk has ~10 M post-net pre-activations, dozens of them within 4e-6 of
        # zero): ONE more oracle backward with every such unit on the other side measures what the ambiguity is worth per tensor, and twice that
        # (the units' contributions may partly cancel in the all-flipped sum) is added to the tensor's bound
        try:
            for (b, key, idx), v in zip(units, saved):
                caches[b][key][idx] = -np.sign(v) * kink_eps if v != 0 else kink_eps
            ogf = TO.backward(cfg, flat, caches, dl)
        finally:
            for (b, key, idx), v in zip(units, saved):
                caches[b][key][idx] = v
        scale = np.abs(og).max()
        for k, (o, shp) in offs.items():
            n = int(np.prod(shp))
            a, r = grad[o:o + n], og[o:o + n]
            bound = a_scale * scale + a_rel * np.abs(r).max() + 2.0 * np.abs(ogf[o:o + n] - r).max()
            assert np.abs(a - r).max() <= bound, "grad mismatch in %s: err %.3e > bound %.3e (incl. the allowance for %d near-kink units)" % (k, np.abs(a - r).max(), bound, len(units))
        return og
    try:
        for signs in itertools.product((1.0, -1.0), repeat=len(units)):
            for (b, key, idx), sg in zip(units, signs):
                caches[b][key][idx] = sg * kink_eps
            ogv = TO.backward(cfg, flat, caches, dl)
            if worst(ogv) is None:
                return ogv
    finally:
        for (b, key, idx), v in zip(units, saved):
            caches[b][key][idx] = v
    raise AssertionError("grad mismatch in %s, for every side of the %d near-kink units too" % (bad[0], len(units)))


def significant_elements(cfg, grads, thr=1e-2):
    """Boolean mask over the flat parameter vector: elements whose gradient is at least `thr` of their tensor's largest at EVERY step of `grads`
    (a list of flat reference gradients, one per Adam step).  Adam moves such an element by lr * m / sqrt(v) with a relative error of the order of
    the gradient's: two correct float32 runs agree on it tightly, whereas an element whose gradient is at noise level may move by a whole step in
    either direction (m / sqrt(v) = +-1 whatever the size of g)."""
    offs, _ = cfg.param_offsets()
    mask = np.ones(grads[0].size, dtype=bool)
    for g in grads:
        for k, (o, shp) in offs.items():
            n = int(np.prod(shp))
            a = np.abs(g[o:o + n])
            mask[o:o + n] &= a >= thr * max(float(a.max()), 1e-30)
    return mask


def assert_weights_after_adam(w, w_ref, lr, steps, tight=2e-6, frac=0.06, far=0.5, significant=None, sig_max=2e-6, sig_frac=0.0):
    """Final weights after `steps` Adam steps against a reference run.  Adam moves an element by ~lr per step whatever its gradient's size
    (m / sqrt(v)), so where a gradient is at fp32-noise level -- e.g. after a post-net pre-activation fell on the other side of a ReLU kink that lies
    within reassociation error of zero (util.assert_grads_match_oracle) -- two correct runs may disagree on a fraction of a step: all but `frac` of the
    elements agree to `tight`, none differs by more than `far` of the distance travelled.  (The per-step LOSS, north_star's criterion, is checked
    to 1e-4 next to this; gradients are compared with the oracle tensor by tensor in the autograd tests.)
    significant: mask of the elements whose reference gradient is well above noise at every step (significant_elements) -- those keep the
    tight bound of the original fixture check: none further than `sig_max` (all but `sig_frac` of them within `tight` when sig_frac > 0), and they
    must be at least a third of the elements, so that the tight check covers the bulk of the model."""
    import numpy as np
    d = np.abs(np.asarray(w, dtype=np.float64) - np.asarray(w_ref, dtype=np.float64))
    assert d.max() <= far * lr * steps, "max |dw| %.3e > %.3e" % (d.max(), far * lr * steps)
    assert (d > tight).mean() < frac, "%.2f %% of the elements differ by more than %.0e" % (100 * (d > tight).mean(), tight)
    if significant is not None:
        assert significant.shape == d.shape and significant.mean() >= 1.0 / 3, "only %.1f %% of the elements count as significant" % (100 * significant.mean())
        ds = d[significant]
        assert ds.max() <= sig_max, "an element with a significant gradient at every step differs by %.3e > %.1e" % (ds.max(), sig_max)
        if sig_frac > 0:
            assert (ds > tight).mean() <= sig_frac, "%.3f %% of the significant elements differ by more than %.0e" % (100 * (ds > tight).mean(), tight)
