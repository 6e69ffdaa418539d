"""GPU parity of the training path: HIP forward / backward / CE / Adam vs the numpy oracle
(itself pinned to the reference's autograd by tests/golden/train.npz) and vs the fixtures."""
import numpy as np
import pytest

from cases import FORWARD_CASES, TRAIN_CASES
from qpnet_amd import synth
import util

pytestmark = pytest.mark.gpu


def _to(dev, *arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


@pytest.mark.parametrize("case", FORWARD_CASES, ids=[c[0] for c in FORWARD_CASES])
def test_forward_logits_vs_reference(case, cuda, golden_dir):
    import torch
    name, cfg, wseed, dseed, bl, ml = case
    g = np.load(golden_dir + "/forward.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda)
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, ml)
    xt, ht, dt, bt = _to(cuda, x, h, d, b)
    with torch.no_grad():
        logits = m(xt, ht, dt, bt)
    lg = logits.cpu().numpy()
    ref = g[name + "_logits"]
    assert lg.shape == ref.shape
    np.testing.assert_allclose(lg, ref, atol=2e-5, rtol=0)      # fp32 reassociation only
    BL = int(b[0])
    lse = np.log(np.exp(lg[0].astype(np.float64)).sum(1))
    loss = (lse - lg[0][np.arange(BL), t[0, -BL:]]).mean()
    assert abs(loss - float(g[name + "_loss"])) < 1e-4          # north_star tolerance


@pytest.mark.parametrize("case", TRAIN_CASES, ids=[c[0] for c in TRAIN_CASES])
def test_autograd_grads_vs_reference(case, cuda, golden_dir):
    """loss.backward() through the HIP backward == the reference's autograd grads (fixture)
    and == the numpy oracle's hand-derived grads, per parameter tensor."""
    import torch
    from oracle import train_oracle as TO
    name, cfg, wseed, dseed, bl, nsteps = case
    g = np.load(golden_dir + "/train.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda).train()
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, 30000)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    BL = int(b[0])
    logits = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    assert abs(loss.item() - g[name + "_losses"][0]) < 1e-4
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    # oracle grads on the same inputs
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    _, dl = TO.ce_loss(lg, t[:, -BL:])
    og0 = TO.backward(cfg, flat, caches, dl)
    og = util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad)
    # the reference's own autograd gradient (fixture; the oracle's default sides are pinned to it by tests/test_oracle_golden.py): where a near-kink
    # unit fell on the other side here (util.assert_grads_match_oracle), the fixture is moved by the oracle's difference between the two sides
    if name + "_grad0" in g:
        ref = g[name + "_grad0"] + (og - og0); mine = grad
    else:
        ref = g[name + "_grad0_sample"] + (og - og0)[::97]; mine = grad[::97]
    assert np.abs(mine - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("case", TRAIN_CASES, ids=[c[0] for c in TRAIN_CASES])
def test_fused_train_steps_vs_reference(case, cuda, golden_dir):
    """forward + CE + backward + Adam entirely behind the C ABI: loss per step within 1e-4 of the
    reference run with the same seed (north_star), final weights match."""
    import torch
    from qpnet_amd.train import FusedTrainer
    name, cfg, wseed, dseed, bl, nsteps = case
    g = np.load(golden_dir + "/train.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    losses = []
    for step in range(nsteps):
        x, h, t, d, b = synth.train_inputs(cfg, bl, dseed + step, 30000)
        xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
        losses.append(tr.step(xt, ht, tt, dt, bt))
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)
    w = m.flat_parameters().cpu().numpy()
    # final weights vs the reference run's (fixture): of the elements whose gradient is significant at every step (numpy oracle on the same chunks) at most 5 %
    # beyond 2e-6 -- the original fixture bound -- and NONE beyond a tenth of a step per step (measured: 3 % and 0.06; a post-net unit on the other side of a
    # ReLU kink than the reference's moves every upstream gradient by 1e-4 .. 1e-3 of its tensor's largest, i.e. by several per cent where the gradient is
    # a hundredth of the largest: tests/f64_child.py shows the sides are the whole difference); the noise-level rest statistically, whole steps allowed
    sig = util.significant_elements(cfg, _oracle_step_grads(cfg, flat, dseed, bl, nsteps))
    util.assert_weights_after_adam(w[::97], g[name + "_wfinal_sample"], 1e-4, nsteps, significant=sig[::97], sig_max=0.1 * 1e-4 * nsteps, sig_frac=0.05)


def _oracle_step_grads(cfg, flat, dseed, bl, nsteps, ml=30000):
    from oracle import train_oracle as TO
    w = flat.copy()
    opt = TO.Adam(w.size)
    return [TO.train_step(cfg, w, opt, *synth.train_inputs(cfg, bl, dseed + step, ml))[1] for step in range(nsteps)]


def test_lagged_loss_is_every_steps_loss_one_step_late(cuda, golden_dir):
    """step(want_loss="lagged") -- what run_train does -- returns the PREVIOUS step's loss without draining the stream, flush_loss() the last
    one: together exactly the losses of step(want_loss=True), i.e. the reference run's (same fixture), each delivered once."""
    from qpnet_amd.train import FusedTrainer
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES[0]
    g = np.load(golden_dir + "/train.npz")
    m = util.build_model(cfg, synth.make_weights(cfg, wseed), cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    got = []
    for step in range(nsteps):
        x, h, t, d, b = synth.train_inputs(cfg, bl, dseed + step, 30000)
        v = tr.step(*_to(cuda, x, h, t, d, b), want_loss="lagged")
        assert (v is None) == (step == 0 or step == 2)              # nothing before the first step, nothing right after a flush
        if v is not None:
            got.append(v)
        if step == 1:                                               # a reporting boundary in the middle
            got.append(tr.flush_loss())
            assert tr.flush_loss() is None                          # delivered once
    got.append(tr.flush_loss())
    assert len(got) == nsteps
    np.testing.assert_allclose(got, g[name + "_losses"], atol=1e-4, rtol=0)
    w = m.flat_parameters().cpu().numpy()
    sig = util.significant_elements(cfg, _oracle_step_grads(cfg, synth.make_weights(cfg, wseed), dseed, bl, nsteps))
    util.assert_weights_after_adam(w[::97], g[name + "_wfinal_sample"], 1e-4, nsteps, significant=sig[::97], sig_max=0.1 * 1e-4 * nsteps, sig_frac=0.05)


def test_torch_adam_on_views_matches(cuda, golden_dir):
    """the drop-in path the reference trainer uses: torch.optim.Adam over model.parameters()
    (which are views of the flat buffer) + our autograd.Function."""
    import torch
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES[0]
    g = np.load(golden_dir + "/train.npz")
    m = util.build_model(cfg, synth.make_weights(cfg, wseed), cuda).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=0.0)
    crit = torch.nn.CrossEntropyLoss()
    losses = []
    for step in range(nsteps):
        x, h, t, d, b = synth.train_inputs(cfg, bl, dseed + step, 30000)
        xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
        BL = int(b[0])
        loss = crit(m(xt, ht, dt, bt).reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)


def test_batch2_grads_specialised_and_generic_wgrad(cuda, monkeypatch):
    """Two chunks per step (rows of both batch items flattened into the weight-gradient contraction: stages that straddle the two items take
    the kernels' generic staging path): the compile-time-tiled k_wgrad3, the generic k_wgrad2 (QPN_WGRAD_GENERIC=1) and the GEMM path's
    k_gemm_tn (QPN_TRAIN_GEMM=1) all match the numpy oracle."""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 21)
    x, h, t, d, b = synth.train_inputs(cfg, 500, 61, 2500)
    xs = np.random.RandomState(5).randint(0, cfg.n_quantize, size=x.shape[1] + 1).astype(np.int64)
    # second row: same features (rows of a batch share the chunk geometry), another waveform
    x = np.stack([x[0], xs[:-1]]); t = np.stack([t[0], xs[1:]])
    h = np.concatenate([h, h]); d = np.concatenate([d, d]); b = np.concatenate([b, b])
    BL = int(b[0])
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    _, dl = TO.ce_loss(lg, t[:, -BL:])
    og = TO.backward(cfg, flat, caches, dl)
    grads = []
    for variant in ("tile", "generic", "gemm"):        # gemm: the LDS-tiled GEMM path (k_gemm_nn / k_gemm_tn) forced on this geometry
        if variant == "generic":
            monkeypatch.setenv("QPN_WGRAD_GENERIC", "1")
        if variant == "gemm":
            monkeypatch.delenv("QPN_WGRAD_GENERIC"); monkeypatch.setenv("QPN_TRAIN_GEMM", "1")
        m = util.build_model(cfg, flat, cuda).train()
        xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
        logits = m(xt, ht, dt, bt)
        loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
        loss.backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy())
    scale = np.abs(og).max()
    for grad in grads:
        util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad)
    # the two tile kernels share everything else; float atomics (upsampling / scatter grads) make runs differ in the last bits
    assert np.abs(grads[0] - grads[1]).max() <= 1e-5 * scale


@pytest.mark.parametrize("knobs", [{"QPN_TRAIN_SERIAL": "1"}, {"QPN_UP_SIDE": "0"}, {"QPN_REDUCE_EARLY": "0"}, {"QPN_POST_WGRAD_PAIR": "0"},
                                   {"QPN_ZERO_IN_POST": "0"}, {"QPN_WR_SIDE": "0"}, {"QPN_WGRAD_CHUNKS_SIDE": "64"}, {"QPN_WGRAD_CHUNKS": "48", "QPN_WGRAD_CHUNKS_SIDE": "32"}, {"QPN_EVENT_FENCE": "1"},
                                   {"QPN_STACK_QUEUE_BWD": "0"}, {"QPN_STACK_QUEUE": "0"}, {"QPN_STACK_WGS_BWD": "512"}, {"QPN_STACK_WGS": "96", "QPN_STACK_WGS_BWD": "40"},
                                   {"QPN_AUX_HOIST": "0"}, {"QPN_AUX_HOIST": "0", "QPN_STACK_QUEUE": "0"}, {"QPN_LAYER_PERSIST": "0", "QPN_LAYER_BWD_PERSIST": "0"},
                                   {"QPN_STACK_WAVE_BWD": "1"}, {"QPN_STACK_WAVE_FWD": "1"}, {"QPN_STACK_WAVE_FWD": "2"}, {"QPN_STACK_WAVE_FWD": "1", "QPN_STACK_WAVE_BWD": "1", "QPN_STACK_WGS": "96", "QPN_STACK_WGS_BWD": "40"}],
                         ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_backward_launch_arrangements_agree(knobs, cuda, monkeypatch):
    """Where the backward's launches run (side stream or not, early reduction, paired post-net contraction, zeroing inside k_post_bwd_w, the
    residual stack as one work-queue launch per direction or a launch per layer, the queues' grid sizes, the auxiliary 1x1 at frame or at
    sample rate ...) is a set of environment knobs, read once when a module's native training state is created: every arrangement yields the
    default one's gradient up to the order of float atomics (and, for QPN_AUX_HOIST, of one fp32 reassociation)."""
    import torch
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 23)
    x, h, t, d, b = synth.train_inputs(cfg, 3000, 67, 30000)
    BL = int(b[0])

    def grad():
        m = util.build_model(cfg, flat, cuda).train()          # a fresh native handle: QPN_EVENT_FENCE is read when its events are created
        xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
        logits = m(xt, ht, dt, bt)
        loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
        loss.backward()
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy(), float(loss)

    g0, l0 = grad()
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g1, l1 = grad()
    assert abs(l0 - l1) < 1e-6
    assert np.abs(g0 - g1).max() <= 1e-5 * np.abs(g0).max()


def test_full_size_step_vs_oracle(cuda):
    """BASELINE config[2] size (paper-size model, batch_length 20000 -> one chunk of ~20.7 k samples): loss within the
    north_star tolerance (1e-4) and every gradient tensor against the numpy oracle's hand-derived backward."""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    x, h, t, d, b = synth.train_inputs(cfg, 20000, 5000, 30000, f0_lo=55.0, f0_hi=300.0)
    BL = int(b[0])
    assert x.shape[1] > 20000
    m = util.build_model(cfg, flat, cuda).train()
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    logits = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    oloss, dl = TO.ce_loss(lg, t[:, -BL:])
    assert abs(loss.item() - float(oloss)) < 1e-4
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lg, atol=5e-5, rtol=0)
    # 20 k-term fp32 sums in two different orders (64 MFMA time chunks vs numpy): allow 2e-4 of the global gradient scale
    util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-4, a_rel=2e-3)


def test_bench_shape_step_vs_oracle(cuda):
    """The exact chunk bench.py times (BASELINE config[1] / SURVEY 8d: batch_length 20000, max_length 30000, the corpus' pitch floor 45 Hz pinned in the
    chunk -> ceil(max d) 62, RF 946, 20 900 samples, 19 954 output rows), default launches (work-queue stack, aux 1x1 at frame rate, side stream):
    loss within the north_star tolerance and every gradient tensor against the numpy oracle."""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    x, h, t, d, b = synth.train_inputs(cfg, 20000, 5000, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True)
    BL = int(b[0])
    assert x.shape[1] == 20900 and BL == 19954 and int(np.ceil(d).max()) == 62
    m = util.build_model(cfg, flat, cuda).train()
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    logits = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    oloss, dl = TO.ce_loss(lg, t[:, -BL:])
    assert abs(loss.item() - float(oloss)) < 1e-4
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lg, atol=5e-5, rtol=0)
    util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-4, a_rel=2e-3)


def _bench_chunk(i=0):
    """bench.py's chunk i on rank 0 of 1 (run_train: seeds 5000 + 17 i): BASELINE config[1] / SURVEY 8d -- 20 900 samples, RF 946, 19 954 output rows."""
    from qpnet_amd.config import PAPER
    x, h, t, d, b = synth.train_inputs(PAPER, 20000, 5000 + 17 * i, 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True)
    assert x.shape[1] == 20900 and int(b[0]) == 19954 and int(np.ceil(d).max()) == 62
    return x, h, t, d, b


def test_the_fused_step_bench_times_vs_oracle(cuda):
    """the path bench.py TIMES -- FusedTrainer.step -> qpn_train_step -> k_post_fb_w<5> (forward, cross entropy and backward of a post-net
    tile as one kernel), both stack work queues, the side stream, the library's Adam -- on the exact bench chunk with default knobs: the loss within the
    north_star tolerance, every tensor of the step's gradient (tr.g) against the numpy oracle's hand-derived backward, the parameters after the step
    against the oracle's Adam (reference loop body: src/bin/qpnet_train.py:517-531)."""
    from oracle import train_oracle as TO
    from qpnet_amd.config import PAPER
    from qpnet_amd.train import FusedTrainer
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    x, h, t, d, b = _bench_chunk(0)
    BL = int(b[0])
    m = util.build_model(cfg, flat, cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    xt, ht, tt, dt = _to(cuda, x, h, t, d)
    assert tr.step(xt, ht, tt, dt, b, want_loss="lagged", maxd=62) is None      # (bench.py: want_loss=False; "lagged" runs the same kernels and keeps the loss)
    loss = tr.flush_loss()
    tr.check_status()
    grad = tr.g[:flat.size].cpu().numpy()
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    oloss, dl = TO.ce_loss(lg, t[:, -BL:])
    assert abs(loss - float(oloss)) < 1e-4
    og = util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-4, a_rel=2e-3)
    wo = flat.copy()
    TO.Adam(flat.size).step(wo, og)
    # (first Adam step: every element moves by lr * g / (|g| + eps) = +-lr -- the sign of a noise-level gradient is noise, hence far = 2; an element
    #  whose gradient is significant moves identically)
    util.assert_weights_after_adam(m.flat_parameters().cpu().numpy(), wo, 1e-4, 1, far=2.0, significant=util.significant_elements(cfg, [og]), sig_max=1e-6)
    # ... and with the loss read inside the step (mode 2 of qpn_train_step: the reference's literal order) the same loss from the same weights
    m2 = util.build_model(cfg, flat, cuda).train()
    assert abs(FusedTrainer(m2, lr=1e-4).step(xt, ht, tt, dt, b, want_loss=True, maxd=62) - loss) < 1e-6


def test_three_fused_steps_at_the_bench_shape_vs_the_torch_port(cuda):
    """three consecutive fused steps on bench.py's first three chunks against oracle/train_torch.py (torch autograd + torch.optim.Adam on
    the CPU, pinned to the reference's fixture): every step's loss within 1e-4 (north_star), the weights after the three Adam steps."""
    from oracle import train_torch as TT
    from qpnet_amd.config import PAPER
    from qpnet_amd.train import FusedTrainer
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    m = util.build_model(cfg, flat, cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    ref = TT.Trainer(cfg, flat, lr=1e-4)
    got, want, grads = [], [], []
    for i in range(3):
        x, h, t, d, b = _bench_chunk(i)
        v = tr.step(*_to(cuda, x, h, t, d), b, want_loss="lagged", maxd=62)
        if v is not None:
            got.append(v)
        loss, grad = ref.loss_and_grad(x, h, t, d, b)
        want.append(loss); grads.append(grad.copy())
        ref.opt.step()
    got.append(tr.flush_loss())
    tr.check_status()
    np.testing.assert_allclose(got, want, atol=1e-4, rtol=0)
    # Measured (tools/adam_parity_stats.py, this shape): the elements with a significant gradient at all three steps (78 % of the model) end within 1.3e-5
    # (0.04 of the 3e-4 travelled), 0.5 % of them beyond 2e-6 -- at this chunk size a post-net unit or two fall on the other side of a ReLU kink
    # (tests/f64_child.py), which moves every upstream gradient by ~1e-4 of its size; noise-level elements may go the other way for whole steps
    util.assert_weights_after_adam(m.flat_parameters().cpu().numpy(), ref.flat.detach().numpy(), 1e-4, 3, far=2.0,
                                   significant=util.significant_elements(cfg, grads), sig_max=3e-5, sig_frac=0.02)


def test_full_size_gradient_error_is_fp32_reassociation_vs_a_float64_oracle(cuda):
    """the full-size gradient tolerances (a_scale 2e-4, a_rel 2e-3, an allowance for ReLU kinks) are loose -- shown here by measurement to be
    about the KINKS, not about the kernels' arithmetic: the oracle run in float64 (train_oracle.precision) is the yardstick, evaluated on the ReLU sides each
    float32 forward actually took (the GPU's are read through the -DQPN_TESTING build's qpn_test_postnet_activations, hence the child process:
    tests/f64_child.py), and per parameter tensor the GPU's gradient -- the fused step bench.py times AND the autograd path -- is no further from it than
    4 x the float32 numpy oracle's own distance (+ 1e-6 of the largest gradient); every unit whose side differs from the float64 run's has a float64
    pre-activation below 1e-5, i.e. the sides differ by rounding alone."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "qpnet_amd", "libqpnet_hip_testing.so")
    assert os.path.exists(lib), "build the testing library first: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, QPN_LIB=lib, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "f64_child.py"), "bench"], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "F64_CHILD_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]


def test_a_flagged_step_leaves_parameters_and_moments_alone(cuda):
    """the device-side status word (here: a target outside [0, n_quantize)) reaches the host up to two steps late in the lagged loop.  The
    Adam kernel reads the word itself and skips the update while it is set: the flagged step and the steps enqueued behind it change neither the
    weights nor the moments, check_status() raises, and training goes on from the last clean state."""
    import torch
    from qpnet_amd import _lib
    from qpnet_amd.config import TINY
    from qpnet_amd.train import FusedTrainer
    cfg = TINY
    m = util.build_model(cfg, synth.make_weights(cfg, 11), cuda).train()
    tr = FusedTrainer(m, lr=1e-3)
    x, h, t, d, b = synth.train_inputs(cfg, 600, 41, 30000)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    maxd = int(np.ceil(d).max())
    tr.step(xt, ht, tt, dt, b, want_loss="lagged", maxd=maxd)
    tr.flush_loss(); tr.check_status()
    w0, m0, v0 = m.flat_parameters().clone(), tr.m.clone(), tr.v.clone()
    bad = tt.clone(); bad[0, -5] = cfg.n_quantize + 3
    tr.step(xt, ht, bad, dt, b, want_loss="lagged", maxd=maxd)          # flagged ...
    tr.step(xt, ht, tt, dt, b, want_loss="lagged", maxd=maxd)           # ... and a clean chunk enqueued behind it before the host has looked
    tr.flush_loss()
    with pytest.raises(_lib.QpnError) as e:
        tr.check_status()                                               # (what run_train does before every report / checkpoint / the final model)
    assert e.value.code == -4
    assert torch.equal(m.flat_parameters(), w0) and torch.equal(tr.m, m0) and torch.equal(tr.v, v0)
    # the host's step count (the bias correction's exponent, what checkpoints store) is set back to the updates the device applied
    assert tr.step_count == 1 and tr.state_dict()["state"][0]["step"] == 1
    tr.step(xt, ht, tt, dt, b, want_loss="lagged", maxd=maxd)           # the word has been read: training goes on
    tr.flush_loss(); tr.check_status()
    assert not torch.equal(m.flat_parameters(), w0) and tr.step_count == 2
    # ... exactly as if the two skipped steps had never been issued: a second trainer that runs the two clean steps alone ends bit-identical
    # (same kernels, same step numbers; the stack's float atomics are the only source of run-to-run noise)
    m2 = util.build_model(cfg, synth.make_weights(cfg, 11), cuda).train()
    tr2 = FusedTrainer(m2, lr=1e-3)
    for _ in range(2):
        tr2.step(xt, ht, tt, dt, b, want_loss="lagged", maxd=maxd)
    tr2.flush_loss(); tr2.check_status()
    np.testing.assert_allclose(m.flat_parameters().cpu().numpy(), m2.flat_parameters().cpu().numpy(), rtol=0, atol=2e-6)
    # the same through step()'s own raise (the lagged collect at the start of a step) and through the in-step check (want_loss=True)
    tr2.step(xt, ht, bad, dt, b, want_loss="lagged", maxd=maxd)
    tr2.step(xt, ht, tt, dt, b, want_loss="lagged", maxd=maxd)
    with pytest.raises(_lib.QpnError):
        tr2.step(xt, ht, tt, dt, b, want_loss="lagged", maxd=maxd)     # raised before anything of this step is enqueued
    assert tr2.step_count == 2
    with pytest.raises(_lib.QpnError):
        tr2.step(xt, ht, bad, dt, b, want_loss=True, maxd=maxd)        # flagged and checked in the call: its Adam launch applied nothing
    assert tr2.step_count == 2
    assert tr2.step(xt, ht, tt, dt, b, want_loss=True, maxd=maxd) > 0 and tr2.step_count == 3


@pytest.mark.parametrize("geo", [(64, 128, 3, 2, 2, 1), (128, 128, 2, 1, 2, 1), (96, 256, 2, 1, 1, 1)], ids=["C64-F3x2", "C128", "C96"])
def test_other_geometry_train_vs_oracle(geo, cuda):
    """Geometries outside the BASELINE configs: another skip width and a repeated fixed stack at n_resch 64, and n_resch 96 / 128
    (column-grouped weight gradients on the generic kernel, partial 64-channel block in the causal-table histogram).
    [Data seed 6: with seed 5 one pre-ReLU skip sum of the C=128 case is -9e-8, the two summation orders disagree on its sign
    and that row's ReLU mask flips -- a 4 % change of one gradient column that is fp32 noise at a kink, not a kernel error.]"""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd.config import QPNetConfig
    C, S, fd, fr, ad, ar = geo
    cfg = QPNetConfig(n_resch=C, n_skipch=S, dilationF_depth=fd, dilationF_repeat=fr, dilationA_depth=ad, dilationA_repeat=ar)
    flat = synth.make_weights(cfg, 7)
    x, h, t, d, b = synth.train_inputs(cfg, 600, 6, 4000)
    BL = int(b[0])
    m = util.build_model(cfg, flat, cuda).train()
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    logits = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    oloss, dl = TO.ce_loss(lg, t[:, -BL:])
    assert abs(loss.item() - float(oloss)) < 1e-4
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lg, atol=2e-5, rtol=0)
    util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-5, a_rel=0.0)


@pytest.mark.parametrize("geo", [(64, 512, 256), (64, 384, 1024), (128, 512, 512)], ids=["S512", "S384-Q1024", "C128-S512-Q512"])
def test_wide_post_net_train_vs_oracle(geo, cuda):
    """n_skipch / n_quantize above 256 (the reference takes any, qpnet.py:174-178): the skip / post-net weight gradients go out as blocks of 256 output rows,
    the causal table's gradient -- an LDS histogram when the one-hot contraction's tiles do not fit -- takes fewer channels per pass."""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd.config import QPNetConfig
    C, S, Q = geo
    cfg = QPNetConfig(n_quantize=Q, n_resch=C, n_skipch=S, dilationF_depth=2, dilationF_repeat=1, dilationA_depth=2, dilationA_repeat=1)
    flat = synth.make_weights(cfg, 7)
    x, h, t, d, b = synth.train_inputs(cfg, 500, 6, 4000)
    BL = int(b[0])
    m = util.build_model(cfg, flat, cuda).train()
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    logits = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    oloss, dl = TO.ce_loss(lg, t[:, -BL:])
    assert abs(loss.item() - float(oloss)) < 1e-4
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lg, atol=2e-5, rtol=0)
    util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-5, a_rel=0.0)


def test_backward_follows_flag_contract_over_the_c_abi(cuda):
    """qpn_train_forward_loss(want_logits | QPN_FWD_BACKWARD_FOLLOWS): the post-net's backward may run inside the forward's launches -- and only a backward of THAT
    dL/dlogits buffer, once, may use it.  Through the C ABI: (a) flag + the same buffer == no flag; (b) flag, then a backward with ANOTHER buffer (2 x dL/dlogits): the
    gradient doubles, i.e. the separate kernel ran on what it was given; (c) a repeated backward of the same forward gives the same gradient again; (d) want_logits bit 0
    still decides whether the logits are written."""
    import ctypes as C
    import torch
    from qpnet_amd import _lib
    from qpnet_amd.config import PAPER
    from qpnet_amd.train import ensure_flat
    cfg = PAPER
    m = util.build_model(cfg, synth.make_weights(cfg, 29), cuda)
    x, h, t, d, b = synth.train_inputs(cfg, 900, 78, 3900)
    xt, ht, tt, dt = _to(cuda, x, h, t, d)
    L, hd = m._native(cuda)
    flat = ensure_flat(m, cuda)
    B, T = xt.shape; BL = int(b[0]); Q = cfg.n_quantize
    maxd = int(np.ceil(d).max())
    stream = torch.cuda.current_stream(cuda).cuda_stream
    args = (hd, flat.data_ptr(), B, T, ht.shape[2], dt.shape[1], BL, maxd, xt.data_ptr(), ht.data_ptr(), dt.data_ptr(), tt.data_ptr(), tt.shape[1])
    lg = torch.full((B, BL, Q), 7.0, device=cuda)
    dl = torch.empty((B, BL, Q), device=cuda)
    n = flat.numel()
    g_plain, g_flag, g_other, g_again = (torch.empty(n, device=cuda) for _ in range(4))
    _lib.check(L.qpn_train_forward_loss(*args, lg.data_ptr(), 0, dl.data_ptr(), stream))
    _lib.check(L.qpn_train_backward(hd, dl.data_ptr(), g_plain.data_ptr(), stream))
    assert bool((lg == 7.0).all())                                        # bit 0 clear: the logits buffer is left alone
    dl_ref = dl.clone()
    _lib.check(L.qpn_train_forward_loss(*args, lg.data_ptr(), 2, dl.data_ptr(), stream))
    assert torch.equal(dl, dl_ref) and bool((lg == 7.0).all())
    _lib.check(L.qpn_train_backward(hd, dl.data_ptr(), g_flag.data_ptr(), stream))
    _lib.check(L.qpn_train_backward(hd, dl.data_ptr(), g_again.data_ptr(), stream))          # (c)
    _lib.check(L.qpn_train_forward_loss(*args, lg.data_ptr(), 3, dl.data_ptr(), stream))
    assert not bool((lg == 7.0).any())                                    # (d) bit 0 set: logits written
    dl2 = dl * 2.0
    _lib.check(L.qpn_train_backward(hd, dl2.data_ptr(), g_other.data_ptr(), stream))         # (b)
    _lib.check(L.qpn_train_status(hd, stream))
    tol = 2e-6 * float(g_plain.abs().max())
    assert float((g_flag - g_plain).abs().max()) <= tol
    assert float((g_again - g_plain).abs().max()) <= tol
    assert float((g_other - 2.0 * g_plain).abs().max()) <= 2 * tol


def test_post_net_tile_kernel_fused_equals_separate(cuda, monkeypatch):
    """qpn_train_step runs the post-net's forward, the cross entropy and the backward of a row tile as ONE kernel (k_post_fb_w: dL/dlogits handed over in LDS, the ReLU
    masks as sign bits in registers); QPN_POST_FUSE=0 keeps k_post_fwd_w + k_post_bwd_w.  Same tile arithmetic: loss identical, gradients equal up to the float-atomics
    order of the OTHER kernels; a ragged last tile (rows past the chunk end) and two batch items included."""
    import torch
    from qpnet_amd.config import PAPER
    from qpnet_amd.train import FusedTrainer
    cfg = PAPER
    flat = synth.make_weights(cfg, 17)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("QPN_POST_FUSE", mode)
        out = []
        for bl, B in ((1500, 1), (333, 2)):
            x, h, t, d, b = synth.train_inputs(cfg, bl, 44, 30000)
            if B == 2:
                x = np.concatenate([x, x[:, ::-1]]); t = np.concatenate([t, t[:, ::-1]]); h = np.concatenate([h, h]); d = np.concatenate([d, d]); b = np.concatenate([b, b])
            m = util.build_model(cfg, flat, cuda).train()
            tr = FusedTrainer(m, lr=1e-4)
            xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
            loss = tr.step(xt, ht, tt, dt, b, want_loss=True)
            tr.check_status()
            out.append((loss, tr.g[:flat.size].cpu().numpy(), tr._dlogits.cpu().numpy()))
        res[mode] = out
    for (l1, g1, dl1), (l0, g0, dl0) in zip(res["1"], res["0"]):
        assert abs(l1 - l0) < 1e-9
        np.testing.assert_array_equal(dl1, dl0)                      # dL/dlogits also reaches memory (the post-net weight gradients read it)
        np.testing.assert_allclose(g1, g0, rtol=0, atol=2e-6 * np.abs(g0).max())


def test_flat_adam_matches_torch_adam(cuda, monkeypatch):
    """FlatAdam (one kernel over the flat parameter buffer) == torch.optim.Adam's OWN update (the step hooks off) on the same loop, three steps."""
    import torch
    from qpnet_amd.config import TINY
    from qpnet_amd.train import FlatAdam
    cfg = TINY
    flat = synth.make_weights(cfg, 11)
    ws = []
    monkeypatch.setenv("QPN_DROPIN_FUSED_ADAM", "0")
    for kind in ("torch", "flat"):
        m = util.build_model(cfg, flat, cuda).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3) if kind == "torch" else FlatAdam(m, lr=1e-3)
        for step in range(3):
            x, h, t, d, b = synth.train_inputs(cfg, 700, 80 + step, 30000)
            xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
            out = m(xt, ht, dt, bt)
            loss = torch.nn.CrossEntropyLoss()(out.reshape(-1, cfg.n_quantize), tt[:, -out.shape[1]:].reshape(-1))
            opt.zero_grad()
            loss.backward()
            opt.step()
        ws.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy())
    np.testing.assert_allclose(ws[0], ws[1], atol=2e-6, rtol=0)


def _ref_loop_step(m, opt, cfg, cuda, seed, bl=700, clip=None):
    import torch
    x, h, t, d, b = synth.train_inputs(cfg, bl, seed, 30000)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    out = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(out.reshape(-1, cfg.n_quantize), tt[:, -out.shape[1]:].reshape(-1))
    opt.zero_grad()
    loss.backward()
    if clip is not None:
        torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
    opt.step()
    return float(loss.item())


def test_stock_adam_stepped_by_the_library_equals_torchs_own(cuda, monkeypatch):
    """The reference loop's `torch.optim.Adam(model.parameters())` (src/bin/qpnet_train.py:426-429) is stepped by the library's Adam kernel through
    the optimizer step hooks (train._adam_prehook).  Against torch's own implementation (hooks off) on the same chunks: weights, moments and step
    counters agree to fp32 rounding, with an lr scheduler, weight decay and in-place gradient clipping in the loop; state_dict() keeps torch's
    layout in both directions (torch -> adopted mid-run, adopted -> torch); a parameter without a gradient hands that step back to torch."""
    import torch
    from qpnet_amd.config import TINY
    cfg = TINY
    flat = synth.make_weights(cfg, 12)

    def run(mode, nsteps, resume=None, drop_grad_at=None):
        monkeypatch.setenv("QPN_DROPIN_FUSED_ADAM", mode)
        m = util.build_model(cfg, flat, cuda).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-3)
        first = 0
        if resume is not None:
            m.load_state_dict(resume["model"]); opt.load_state_dict(resume["opt"]); first = resume["step"]
        n_params = len(opt.param_groups[0]["params"])
        for step in range(first, nsteps):
            opt.param_groups[0]["lr"] = 1e-3 * 0.5 ** (step // 2)          # (what a StepLR(step_size=2, gamma=0.5) writes into the group)
            if step == drop_grad_at:
                # torch skips a parameter whose gradient is None (no moment decay, no step count): that step is torch's, the next ones the library's again
                x, h, t, d, b = synth.train_inputs(cfg, 700, 80 + step, 30000)
                xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
                out = m(xt, ht, dt, bt)
                loss = torch.nn.CrossEntropyLoss()(out.reshape(-1, cfg.n_quantize), tt[:, -out.shape[1]:].reshape(-1))
                opt.zero_grad(); loss.backward()
                m.causal.conv.bias.grad = None
                opt.step()
            else:
                _ref_loop_step(m, opt, cfg, cuda, 80 + step, clip=5.0)
            assert len(opt.param_groups[0]["params"]) == n_params          # (emptied only while torch's step body runs)
        return m, opt

    m0, o0 = run("0", 5)
    m1, o1 = run("1", 5)
    assert o1.__dict__["_qpn_adopt"] and not o0.__dict__.get("_qpn_adopt")
    w0 = torch.cat([p.detach().reshape(-1) for p in m0.parameters()]).cpu().numpy()
    w1 = torch.cat([p.detach().reshape(-1) for p in m1.parameters()]).cpu().numpy()
    np.testing.assert_allclose(w1, w0, atol=2e-6, rtol=0)
    assert np.abs(w0 - flat).max() > 1e-3
    s0, s1 = o0.state_dict(), o1.state_dict()
    assert s0["param_groups"][0]["params"] == s1["param_groups"][0]["params"] and set(s0["state"]) == set(s1["state"])
    assert abs(s1["param_groups"][0]["lr"] - s0["param_groups"][0]["lr"]) < 1e-12 and s1["param_groups"][0]["lr"] < 1e-3     # the scheduler's lr is what stepped
    for i in s0["state"]:
        assert float(s1["state"][i]["step"]) == float(s0["state"][i]["step"]) == 5.0
        for k in ("exp_avg", "exp_avg_sq"):
            a, b = s0["state"][i][k].cpu().numpy(), s1["state"][i][k].cpu().numpy()
            assert a.shape == b.shape
            # (two runs of the same backward differ by float-atomics order, ~1e-6 of the largest gradient; the second moment squares it)
            np.testing.assert_allclose(b, a, rtol=0, atol=(1e-5 if k == "exp_avg" else 4e-5) * max(float(np.abs(a).max()), 1e-30))
    # resume across the two implementations, both ways, at step 3 of 5
    ma, oa = run("0", 3)
    ckpt = {"model": {k: v.clone() for k, v in ma.state_dict().items()}, "opt": oa.state_dict(), "step": 3}
    m2, o2 = run("1", 5, resume=ckpt)                    # torch-made state, continued by the library
    assert o2.__dict__["_qpn_adopt"]
    mb, ob = run("1", 3)
    ckpt = {"model": {k: v.clone() for k, v in mb.state_dict().items()}, "opt": ob.state_dict(), "step": 3}
    m3, o3 = run("0", 5, resume=ckpt)                    # library-made state, continued by torch
    for mm in (m2, m3):
        w = torch.cat([p.detach().reshape(-1) for p in mm.parameters()]).cpu().numpy()
        np.testing.assert_allclose(w, w0, atol=3e-6, rtol=0)
    # a missing gradient: identical to torch's own handling of the same loop
    m4, o4 = run("0", 5, drop_grad_at=2)
    m5, o5 = run("1", 5, drop_grad_at=2)
    w4 = torch.cat([p.detach().reshape(-1) for p in m4.parameters()]).cpu().numpy()
    w5 = torch.cat([p.detach().reshape(-1) for p in m5.parameters()]).cpu().numpy()
    np.testing.assert_allclose(w5, w4, atol=3e-6, rtol=0)
    assert o5.__dict__["_qpn_adopt"] is False            # (step counts differ across parameters from there on: torch's loop keeps the optimizer)
    s4, s5 = o4.state_dict(), o5.state_dict()
    assert [float(s5["state"][i]["step"]) for i in s5["state"]] == [float(s4["state"][i]["step"]) for i in s4["state"]]


def test_stock_adam_rollback_keeps_the_loaded_step_counts(cuda, monkeypatch):
    """`opt.load_state_dict()` on an optimizer the step hooks have adopted -- a rollback to an earlier checkpoint after stepping on, or an empty
    state -- must continue from the LOADED state (its step counts drive the bias correction), not from the adopter's old count: 2 steps, checkpoint,
    2 more steps, rollback, 2 steps == 4 steps straight through, the counters read 4, and a freshly made (empty) state loads and steps without a KeyError."""
    import copy
    import torch
    from qpnet_amd.config import TINY
    cfg = TINY
    flat = synth.make_weights(cfg, 12)
    monkeypatch.setenv("QPN_DROPIN_FUSED_ADAM", "1")

    def weights(m):
        return torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy()

    m = util.build_model(cfg, flat, cuda).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    empty = copy.deepcopy(opt.state_dict())
    for step in range(4):
        _ref_loop_step(m, opt, cfg, cuda, 90 + step)
    w_straight = weights(m)
    m = util.build_model(cfg, flat, cuda).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for step in range(2):
        _ref_loop_step(m, opt, cfg, cuda, 90 + step)
    ckpt = {"model": {k: v.clone() for k, v in m.state_dict().items()}, "opt": copy.deepcopy(opt.state_dict())}
    for step in range(2, 4):
        _ref_loop_step(m, opt, cfg, cuda, 90 + step)
    assert opt.__dict__["_qpn_adopt"] and opt.__dict__["_qpn_adopt"].steps == 4
    m.load_state_dict(ckpt["model"]); opt.load_state_dict(ckpt["opt"])            # roll back to step 2
    for step in range(2, 4):
        _ref_loop_step(m, opt, cfg, cuda, 90 + step)
    np.testing.assert_allclose(weights(m), w_straight, atol=3e-6, rtol=0)
    sd = opt.state_dict()
    assert [float(st["step"]) for st in sd["state"].values()] == [4.0] * len(sd["state"])
    assert opt.__dict__["_qpn_adopt"]                                             # (adopted again after the step torch ran itself)
    # an empty state (a fresh optimizer's) over a stepped one: starts over from step 0
    m.load_state_dict(ckpt["model"]); opt.load_state_dict(empty)
    for step in range(2):
        _ref_loop_step(m, opt, cfg, cuda, 90 + step)
    sd = opt.state_dict()
    assert [float(st["step"]) for st in sd["state"].values()] == [2.0] * len(sd["state"])
    assert all(set(st) == {"step", "exp_avg", "exp_avg_sq"} for st in sd["state"].values())


def test_stock_adam_left_alone_when_not_this_modules(cuda, monkeypatch):
    """two groups, or a closure: torch's own step runs (with fused=True on eligible groups), results as before."""
    import torch
    from qpnet_amd.config import TINY
    cfg = TINY
    monkeypatch.setenv("QPN_DROPIN_FUSED_ADAM", "1")
    m = util.build_model(cfg, synth.make_weights(cfg, 12), cuda).train()
    ps = list(m.parameters())
    opt = torch.optim.Adam([{"params": ps[:4]}, {"params": ps[4:], "lr": 5e-4}], lr=1e-3)
    _ref_loop_step(m, opt, cfg, cuda, 80)
    assert opt.__dict__["_qpn_adopt"] is False and all(g["fused"] for g in opt.param_groups)
    opt2 = torch.optim.Adam(m.parameters(), lr=1e-3)
    x, h, t, d, b = synth.train_inputs(cfg, 700, 81, 30000)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)

    def closure():
        opt2.zero_grad()
        out = m(xt, ht, dt, bt)
        loss = torch.nn.CrossEntropyLoss()(out.reshape(-1, cfg.n_quantize), tt[:, -out.shape[1]:].reshape(-1))
        loss.backward()
        return loss
    w_before = m.flat_parameters().clone()
    loss = opt2.step(closure)
    assert float(loss.detach()) > 0 and float((m.flat_parameters().detach() - w_before).abs().max()) > 1e-5
    assert not opt2.__dict__.get("_qpn_adopt")


def test_an_interior_parameter_replaced_is_picked_up_within_a_few_forwards(cuda):
    """ensure_flat's per-call check looks at the first, the last and four rotating parameter views (the full walk runs every 32nd call).  One interior
    `p.data = ...` -- which leaves the flat buffer the kernels read stale -- is noticed within 30 calls: the buffer is rebuilt from the parameters and the
    logits are those of the new value."""
    import torch
    from qpnet_amd.config import TINY
    from qpnet_amd.train import ensure_flat
    cfg = TINY
    flat = synth.make_weights(cfg, 4)
    m = util.build_model(cfg, flat, cuda)
    x, h, t, d, b = synth.train_inputs(cfg, 300, 17, 30000)
    xt, ht, dt, bt = _to(cuda, x, h, d, b)
    with torch.no_grad():
        m(xt, ht, dt, bt); m(xt, ht, dt, bt)
        params = list(m.parameters())
        q = params[len(params) // 2]
        q.data = torch.zeros_like(q.data)                                 # a new storage for ONE interior parameter
        for calls in range(1, 40):
            out = m(xt, ht, dt, bt)
            f = m._flat
            if f.data_ptr() <= q.data_ptr() < f.data_ptr() + 4 * f.numel():
                break
        assert calls <= 31
        m2 = util.build_model(cfg, ensure_flat(m, cuda).cpu().numpy(), cuda)
        assert float(q.abs().max()) == 0.0 and torch.equal(out, m2(xt, ht, dt, bt))


def test_grad_accumulation_and_zero_grad_in_place(cuda):
    """The autograd backward hands out views of a FRESH flat buffer: two backwards before a step accumulate (g1 + g2, not
    2 * g2), and zero_grad(set_to_none=False) followed by a backward gives exactly the new gradient."""
    import torch
    from qpnet_amd.config import TINY
    cfg = TINY
    m = util.build_model(cfg, synth.make_weights(cfg, 21), cuda).train()
    crit = torch.nn.CrossEntropyLoss()

    def loss_of(seed):
        x, h, t, d, b = synth.train_inputs(cfg, 500, seed, 30000)
        xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
        BL = int(b[0])
        return crit(m(xt, ht, dt, bt).reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))

    def flat_grad():
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

    m.zero_grad(set_to_none=True); loss_of(71).backward(); g1 = flat_grad()
    m.zero_grad(set_to_none=True); loss_of(72).backward(); g2 = flat_grad()
    m.zero_grad(set_to_none=True); loss_of(71).backward(); loss_of(72).backward(); acc = flat_grad()
    scale = float(g1.abs().max())
    assert float((acc - (g1 + g2)).abs().max()) <= 1e-5 * scale
    m.zero_grad(set_to_none=False); loss_of(72).backward(); again = flat_grad()
    assert float((again - g2).abs().max()) <= 1e-5 * scale
    # FlatAdam finds the flat buffer behind consecutive p.grad views, and gathers when they are not
    from qpnet_amd.train import _flat_grad_of
    params = list(m.parameters())
    assert _flat_grad_of(params) is not None
    params[3].grad = params[3].grad.clone()
    assert _flat_grad_of(params) is None


def test_anchored_backward_has_loss_backward_semantics(cuda, monkeypatch):
    """The drop-in module's backward when every parameter trains (the reference trainer, qpnet_train.py:527-531): one anchor leaf in the graph, the
    120 gradients assigned by the backward itself as views of ONE persistent buffer (train._anchored_backward).  It must behave like autograd's own
    accumulation: same values as the classic path (QPN_DROPIN_FLAT_GRAD=0), a second backward adds, a gradient somebody replaced is added to in
    place, and a model with a frozen parameter goes the classic way (its frozen parameter gets no gradient)."""
    import torch
    from qpnet_amd.config import TINY
    from qpnet_amd.train import _flat_grad_of
    cfg = TINY
    crit = torch.nn.CrossEntropyLoss()

    def loss_of(m, seed):
        x, h, t, d, b = synth.train_inputs(cfg, 500, seed, 30000)
        xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
        return crit(m(xt, ht, dt, bt).reshape(-1, cfg.n_quantize), tt[:, -int(b[0]):].reshape(-1))

    def flat_grad(m):
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

    monkeypatch.setenv("QPN_DROPIN_FLAT_GRAD", "0")
    mc = util.build_model(cfg, synth.make_weights(cfg, 21), cuda).train()
    loss_of(mc, 71).backward(); c1 = flat_grad(mc)
    mc.zero_grad(set_to_none=True); loss_of(mc, 72).backward(); c2 = flat_grad(mc)
    assert "_qpn_gflat" not in mc.__dict__
    monkeypatch.delenv("QPN_DROPIN_FLAT_GRAD")
    m = util.build_model(cfg, synth.make_weights(cfg, 21), cuda).train()
    scale = float(c1.abs().max())
    loss_of(m, 71).backward()
    params = list(m.parameters())
    buf = _flat_grad_of(params)
    assert buf is not None and buf.data_ptr() == m.__dict__["_qpn_gflat"][1].data_ptr()          # consecutive views of the persistent buffer
    assert float((flat_grad(m) - c1).abs().max()) <= 1e-5 * scale
    loss_of(m, 72).backward()                                                                     # accumulates (one add)
    assert float((flat_grad(m) - (c1 + c2)).abs().max()) <= 1e-5 * scale
    m.zero_grad(set_to_none=True); loss_of(m, 72).backward()                                      # the usual step: written in place, same buffer
    assert _flat_grad_of(params).data_ptr() == buf.data_ptr()
    assert float((flat_grad(m) - c2).abs().max()) <= 1e-5 * scale
    params[3].grad = params[3].grad.clone(); params[5].grad = None                                # somebody touched two of them
    loss_of(m, 71).backward()
    assert float((flat_grad(m) - (c1 + c2)).abs()[: sum(p.numel() for p in params[:5])].max()) <= 1e-5 * scale
    o5 = sum(p.numel() for p in params[:5]); n5 = params[5].numel()
    assert float((params[5].grad.reshape(-1) - c1[o5:o5 + n5]).abs().max()) <= 1e-5 * scale      # (the one that had none holds the new gradient only)
    m.zero_grad(set_to_none=True)
    params[7].requires_grad_(False)                                                                # a frozen parameter: the classic path
    loss_of(m, 72).backward()
    assert params[7].grad is None and params[8].grad is not None
    g = torch.cat([p.grad.reshape(-1) for i, p in enumerate(params) if i != 7])
    o7 = sum(p.numel() for p in params[:7]); n7 = params[7].numel()
    assert float((g - torch.cat([c2[:o7], c2[o7 + n7:]])).abs().max()) <= 1e-5 * scale


def test_backward_of_a_replaced_forward_raises(cuda):
    """one outstanding forward per model: a validation forward between forward and backward must not silently feed the
    wrong activations to the backward kernels"""
    import torch
    from qpnet_amd.config import TINY
    cfg = TINY
    m = util.build_model(cfg, synth.make_weights(cfg, 22), cuda).train()
    x, h, t, d, b = synth.train_inputs(cfg, 400, 81, 30000)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    out = m(xt, ht, dt, bt)
    with torch.no_grad():
        m(xt, ht, dt, bt)                      # e.g. a validation pass
    with pytest.raises(RuntimeError, match="one outstanding forward"):
        out.sum().backward()
    out2 = m(xt, ht, dt, bt)
    out2.sum().backward()                      # the current forward is fine


def test_fused_step_normalises_inputs_and_flags_bad_targets(cuda):
    import torch
    from qpnet_amd.config import TINY
    from qpnet_amd.train import FusedTrainer
    from qpnet_amd import _lib
    cfg = TINY
    flat = synth.make_weights(cfg, 23)
    x, h, t, d, b = synth.train_inputs(cfg, 400, 91, 30000)
    ma = util.build_model(cfg, flat, cuda).train(); mb = util.build_model(cfg, flat, cuda).train()
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    la = FusedTrainer(ma).step(xt, ht, tt, dt, bt)
    # int32 samples, float64 factors, a non-contiguous feature tensor: same step
    hT = ht.transpose(1, 2).contiguous().transpose(1, 2)
    lb = FusedTrainer(mb).step(xt.int(), hT, tt.int(), dt.double(), bt)
    # (the loss sum and the adaptive blocks' scatter-add use float atomics: equal up to summation order)
    assert abs(la - lb) < 1e-9
    assert float((ma.flat_parameters() - mb.flat_parameters()).abs().max()) < 1e-6
    bad = tt.clone(); bad[0, -5] = cfg.n_quantize + 3
    with pytest.raises(_lib.QpnError) as e:
        FusedTrainer(ma).step(xt, ht, bad, dt, bt)
    assert e.value.code == -4


def _same_after_adam(w, w_ref, lr, steps):
    """Adam moves an element by ~lr per step whatever its gradient's size, so where a gradient is at fp32-noise level two
    summation orders (float atomics, torch's kernels) may disagree on a fraction of a step: almost all elements agree to
    2e-6, none differs by more than a fifth of the distance travelled."""
    d = (w - w_ref).abs()
    assert float(d.max()) <= 0.2 * lr * steps
    assert float((d > 2e-6).float().mean()) < 1e-3


def test_trainer_checkpoint_resumes_like_torch_adam(cuda, tmp_path):
    """FusedTrainer state -> checkpoint in the reference's {"model","optimizer","iterations"} format -> (a) a fresh
    FusedTrainer continues identically (up to float-atomics order), (b) torch.optim.Adam on the drop-in module continues within fp32 noise."""
    import torch
    from qpnet_amd import loaders
    from qpnet_amd.config import TINY
    from qpnet_amd.qpnet import QPNet
    from qpnet_amd.train import FusedTrainer
    cfg = TINY
    m = util.build_model(cfg, synth.make_weights(cfg, 24), cuda).train()
    tr = FusedTrainer(m, lr=1e-3)
    data = [_to(cuda, *synth.train_inputs(cfg, 400, 95 + i, 30000)) for i in range(4)]
    for i in range(2):
        tr.step(*data[i])
    path = loaders.save_checkpoint(str(tmp_path), m, tr, 2)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"model", "optimizer", "iterations"} and set(ck["optimizer"]) == {"state", "param_groups"}
    ref_losses = [tr.step(*data[i]) for i in (2, 3)]
    w_ref = m.flat_parameters().clone()
    # (a)
    m2 = QPNet(**cfg.kwargs()); tr2 = FusedTrainer(m2.to(cuda).train())
    assert loaders.load_checkpoint(path, tr2.model, tr2) == 2 and tr2.lr == 1e-3 and tr2.step_count == 2
    np.testing.assert_allclose([tr2.step(*data[i]) for i in (2, 3)], ref_losses, rtol=0, atol=1e-6)
    _same_after_adam(tr2.model.flat_parameters(), w_ref, 1e-3, 2)
    # (b)
    m3 = QPNet(**cfg.kwargs()).to(cuda).train()
    opt = torch.optim.Adam(m3.parameters(), lr=1e-4)
    loaders.load_checkpoint(path, m3, opt)
    crit = torch.nn.CrossEntropyLoss()
    for i in (2, 3):
        xt, ht, tt, dt, bt = data[i]
        BL = int(bt[0])
        loss = crit(m3(xt, ht, dt, bt).reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
        opt.zero_grad(); loss.backward(); opt.step()
    _same_after_adam(m3.flat_parameters(), w_ref, 1e-3, 2)


# ---------------------------------------------------------------- the LDS-tiled GEMM path (train_gemm.hip)
@pytest.mark.parametrize("case", TRAIN_CASES, ids=[c[0] for c in TRAIN_CASES])
def test_gemm_path_on_the_small_geometries(case, cuda, golden_dir, monkeypatch):
    """QPN_TRAIN_GEMM=1 runs the wide-stack kernels on the tiny / paper-size geometries: same reference fixtures
    (loss per step within 1e-4, final weights), same oracle gradients per tensor."""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd.train import FusedTrainer
    monkeypatch.setenv("QPN_TRAIN_GEMM", "1")
    name, cfg, wseed, dseed, bl, nsteps = case
    g = np.load(golden_dir + "/train.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda).train()
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, 30000)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    BL = int(b[0])
    logits = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    assert abs(loss.item() - g[name + "_losses"][0]) < 1e-4
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lg, atol=3e-5, rtol=0)
    _, dl = TO.ce_loss(lg, t[:, -BL:])
    util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-5, a_rel=1e-4)
    m2 = util.build_model(cfg, flat, cuda).train()
    tr = FusedTrainer(m2, lr=1e-4)
    losses = []
    for step in range(nsteps):
        xs = _to(cuda, *synth.train_inputs(cfg, bl, dseed + step, 30000))
        losses.append(tr.step(*xs))
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)
    util.assert_weights_after_adam(m2.flat_parameters().cpu().numpy()[::97], g[name + "_wfinal_sample"], 1e-4, nsteps)


def test_default_geometry_vs_reference(cuda, golden_dir):
    """The geometry runQP.py instantiates (n_resch 512, 12 fixed + 4 adaptive layers; src/utils/param_model.py:58-64):
    forward logits, loss, per-tensor gradients (numpy oracle) and two fused Adam steps against the reference fixture."""
    import torch
    from cases import FORWARD_CASES_D, TRAIN_CASES_D
    from oracle import train_oracle as TO
    from qpnet_amd.train import FusedTrainer
    name, cfg, wseed, dseed, bl, ml = FORWARD_CASES_D[0]
    g = np.load(golden_dir + "/forward_d.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda)
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, ml)
    with torch.no_grad():
        lg = m(*_to(cuda, x, h, d, b)).cpu().numpy()
    np.testing.assert_allclose(lg, g[name + "_logits"], atol=5e-5, rtol=0)
    BL = int(b[0])
    lse = np.log(np.exp(lg[0].astype(np.float64)).sum(1))
    assert abs((lse - lg[0][np.arange(BL), t[0, -BL:]]).mean() - float(g[name + "_loss"])) < 1e-4
    # gradients
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES_D[0]
    g = np.load(golden_dir + "/train_d.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda).train()
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, 2000)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    BL = int(b[0])
    loss = torch.nn.CrossEntropyLoss()(m(xt, ht, dt, bt).reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    assert abs(loss.item() - g[name + "_losses"][0]) < 1e-4
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    ref = g[name + "_grad0_sample"]
    assert np.abs(grad[::97] - ref).max() <= 1e-4 * np.abs(ref).max()
    lgo, caches = TO.forward(cfg, flat, x, h, d, b)
    _, dl = TO.ce_loss(lgo, t[:, -BL:])
    util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-5, a_rel=2e-4)
    # two fused steps
    m2 = util.build_model(cfg, flat, cuda).train()
    tr = FusedTrainer(m2, lr=1e-4)
    losses = [tr.step(*_to(cuda, *synth.train_inputs(cfg, bl, dseed + s, 2000))) for s in range(nsteps)]
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)
    # Adam moves every element by ~lr per step whatever the gradient's size: where a gradient is at fp32-noise level the
    # two summation orders can disagree on a few percent of one step (lr = 1e-4): 5e-6 bounds that, 99.998 % are within 2e-6
    w, wr = m2.flat_parameters().cpu().numpy()[::97], g[name + "_wfinal_sample"]
    np.testing.assert_allclose(w, wr, atol=5e-6, rtol=0)
    assert (np.abs(w - wr) > 2e-6).mean() < 1e-4


@pytest.mark.parametrize("cfgname", ["tiny", "paper"])
def test_fused_forward_loss_equals_forward_then_ce(cfgname, cuda, monkeypatch):
    """qpn_train_forward_loss (cross entropy inside the post-net kernel) against qpn_train_forward + qpn_ce_loss:
    same logits, same dL/dlogits bit for bit (same per-row arithmetic), same loss up to the order of the partial sums;
    with want_logits = 0 the logits buffer is left alone.  (reference qpnet_train.py:520-528)"""
    import ctypes as C
    import torch
    from qpnet_amd import _lib
    from qpnet_amd.config import TINY, PAPER
    from qpnet_amd.train import ensure_flat
    cfg = TINY if cfgname == "tiny" else PAPER
    m = util.build_model(cfg, synth.make_weights(cfg, 23), cuda)
    bl = 700 if cfgname == "tiny" else 1500
    x, h, t, d, b = synth.train_inputs(cfg, bl, 77, bl + 3000)
    x, h, t, d = np.repeat(x, 2, 0), np.repeat(h, 2, 0), np.repeat(t, 2, 0).copy(), np.repeat(d, 2, 0)
    t[1, -5] = (t[1, -5] + 7) % cfg.n_quantize                       # two different rows in the batch
    xt, ht, tt, dt = _to(cuda, x, h, t, d)
    L, hd = m._native(cuda)
    flat = ensure_flat(m, cuda)
    B, T = xt.shape; BL = int(b[0]); Q = cfg.n_quantize
    maxd = int(np.ceil(d).max())
    stream = torch.cuda.current_stream(cuda).cuda_stream
    lg0 = torch.empty((B, BL, Q), device=cuda); dl0 = torch.empty_like(lg0)
    lg1 = torch.full((B, BL, Q), 7.0, device=cuda); dl1 = torch.empty_like(lg0)
    l0, l1 = C.c_double(0), C.c_double(0)
    args = (hd, flat.data_ptr(), B, T, ht.shape[2], dt.shape[1], BL, maxd, xt.data_ptr(), ht.data_ptr(), dt.data_ptr())
    _lib.check(L.qpn_train_forward(*args, lg0.data_ptr(), stream))
    _lib.check(L.qpn_ce_loss(hd, lg0.data_ptr(), tt.data_ptr(), tt.shape[1], B, BL, dl0.data_ptr(), C.byref(l0), stream))
    _lib.check(L.qpn_train_forward_loss(*args, tt.data_ptr(), tt.shape[1], lg1.data_ptr(), 1, dl1.data_ptr(), stream))
    _lib.check(L.qpn_train_loss(hd, C.byref(l1), stream))
    _lib.check(L.qpn_train_status(hd, stream))
    assert torch.equal(lg0, lg1) and torch.equal(dl0, dl1)
    assert abs(l0.value - l1.value) < 1e-12 and l0.value > 0
    # gradient through the fused path == through the separate one
    g0 = torch.empty(flat.numel(), device=cuda); g1 = torch.empty_like(g0)
    _lib.check(L.qpn_train_backward(hd, dl1.data_ptr(), g1.data_ptr(), stream))
    _lib.check(L.qpn_train_forward(*args, lg0.data_ptr(), stream))
    _lib.check(L.qpn_train_backward(hd, dl0.data_ptr(), g0.data_ptr(), stream))
    torch.cuda.synchronize()
    assert float((g0 - g1).abs().max()) <= 1e-6 * max(1.0, float(g0.abs().max()))
    # want_logits = 0: loss and gradient only
    lg2 = torch.full((B, BL, Q), 7.0, device=cuda); dl2 = torch.empty_like(lg0)
    _lib.check(L.qpn_train_forward_loss(*args, tt.data_ptr(), tt.shape[1], lg2.data_ptr(), 0, dl2.data_ptr(), stream))
    _lib.check(L.qpn_train_loss(hd, C.byref(l1), stream))
    assert torch.equal(dl2, dl0) and abs(l0.value - l1.value) < 1e-12
    if cfgname == "paper":
        assert float(lg2.min()) == 7.0                               # fused in the kernel: never written
    # the separate kernel behind the forward gives the same (QPN_CE_SEPARATE: the route wide stacks take)
    monkeypatch.setenv("QPN_CE_SEPARATE", "1")                       # (read when a module's training state is created: a fresh module)
    m2 = util.build_model(cfg, synth.make_weights(cfg, 23), cuda)
    L, hd2 = m2._native(cuda)
    flat2 = ensure_flat(m2, cuda)
    args2 = (hd2, flat2.data_ptr()) + args[2:]
    _lib.check(L.qpn_train_forward_loss(*args2, tt.data_ptr(), tt.shape[1], lg2.data_ptr(), 0, dl2.data_ptr(), stream))
    _lib.check(L.qpn_train_loss(hd2, C.byref(l1), stream))
    assert torch.equal(dl2, dl0) and torch.equal(lg2, lg0) and abs(l0.value - l1.value) < 1e-12


def test_forward_needs_no_maxd_read_back(cuda):
    """QPNet.forward derives maxd from the chunk's SHAPES (train.forward_maxd) instead of reading ceil(max d) back from the
    device.  A chunk that is longer than RF + batch_length (here: a smaller batch_length on the same chunk) makes that bound
    larger than the true ceil(max d): the logits must not change (extra leading context only)."""
    import torch
    from qpnet_amd.config import TINY
    from qpnet_amd.train import forward_maxd
    cfg = TINY
    m = util.build_model(cfg, synth.make_weights(cfg, 11), cuda)
    x, h, t, d, b = synth.train_inputs(cfg, 600, 41, 30000)
    xt, ht, dt, _ = _to(cuda, x, h, d, b)
    b2 = torch.tensor([int(b[0]) - 137])                       # host tensor: no read-back at all
    true_maxd = int(np.ceil(d).max())
    # (synth.train_inputs sizes the chunk for the largest factor of the whole utterance: the bound may already exceed this chunk's own)
    assert forward_maxd(m, x.shape[1], h.shape[2], d.shape[1], int(b[0]), dt) >= true_maxd
    assert forward_maxd(m, x.shape[1], h.shape[2], d.shape[1], int(b2[0]), dt) > true_maxd + 30
    with torch.no_grad():
        lg_bound = m(xt, ht, dt, b2).cpu().numpy()
        m.read_back_maxd = True
        lg_exact = m(xt, ht, dt, b2).cpu().numpy()
        m.read_back_maxd = False
        full = m(xt, ht, dt, torch.from_numpy(b)).cpu().numpy()
    np.testing.assert_allclose(lg_bound, lg_exact, atol=1e-6, rtol=0)
    np.testing.assert_allclose(lg_bound, full[:, 137:], atol=1e-6, rtol=0)      # the last rows of the longer window are the same rows


def test_forward_status_is_raised_before_the_optimizer_step(cuda):
    """The reference asserts on the gather bounds inside forward (qpnet.py:294).  Here the device-side check costs no stream drain in a
    training loop: a forward that records a graph enqueues it, and backward() -- i.e. BEFORE any optimizer step the bad chunk could feed --
    collects it; a forward under torch.no_grad() (nothing else may follow it) checks in the call; model.check_status() collects on demand."""
    import torch
    from qpnet_amd import _lib
    from qpnet_amd.config import TINY
    cfg = TINY
    m = util.build_model(cfg, synth.make_weights(cfg, 11), cuda).train()
    x, h, t, d, b = synth.train_inputs(cfg, 600, 41, 30000)
    xt, ht, dt, bt = _to(cuda, x, h, d, b)
    bad = dt * 40.0                                               # taps far outside the chunk
    with torch.no_grad():
        with pytest.raises(_lib.QpnError) as e:
            m(xt, ht, bad, torch.from_numpy(b))                   # the LAST forward of an eval loop must not go unreported
        assert e.value.code == -4
        lg = m(xt, ht, dt, torch.from_numpy(b))                   # reported once; the module keeps working
    assert np.isfinite(lg.cpu().numpy()).all()
    w0 = m.flat_parameters().clone()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    logits = m(xt, ht, bad, torch.from_numpy(b))                  # training forward: returns, nothing has been read back
    with pytest.raises(_lib.QpnError) as e:
        logits.sum().backward()
    assert e.value.code == -4
    assert torch.equal(m.flat_parameters(), w0)                   # ... and no update was applied
    logits = m(xt, ht, dt, torch.from_numpy(b)); logits.sum().backward(); opt.step()
    m(xt, ht, bad, torch.from_numpy(b))                           # a graph-recording forward with no backward behind it ...
    with pytest.raises(_lib.QpnError):
        m.check_status()                                          # ... is collected on demand
    m.check_status()


def test_fused_step_reports_a_bad_chunk_at_the_next_step(cuda):
    """FusedTrainer.step(want_loss=False) never drains the stream: the status word of step i is copied behind it and raised when step i+2
    starts at the latest (not up to 99 steps late as when it was read every 100 steps); check_status() collects everything outstanding."""
    import torch
    from qpnet_amd import _lib
    from qpnet_amd.config import TINY
    from qpnet_amd.train import FusedTrainer
    cfg = TINY
    m = util.build_model(cfg, synth.make_weights(cfg, 11), cuda).train()
    tr = FusedTrainer(m, lr=1e-4)
    x, h, t, d, b = synth.train_inputs(cfg, 600, 41, 30000)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    maxd = int(np.ceil(d).max())
    tr.step(xt, ht, tt, dt, b, want_loss=False, maxd=maxd)
    tr.step(xt, ht, tt, dt * 40.0, b, want_loss=False, maxd=maxd)          # bad chunk: enqueued, not yet seen
    with pytest.raises(_lib.QpnError) as e:
        tr.step(xt, ht, tt, dt, b, want_loss=False, maxd=maxd)             # (may still be queued on the device ...)
        tr.step(xt, ht, tt, dt, b, want_loss=False, maxd=maxd)             # ... two steps later it has been seen
    assert e.value.code == -4
    tr.step(xt, ht, tt, dt * 40.0, b, want_loss=False, maxd=maxd)
    with pytest.raises(_lib.QpnError):
        tr.check_status()
    tr.check_status()


def _u0_inputs(cfg_u, bl, seed, ml):
    """a chunk for an upsampling_factor = 0 model: the features arrive at SAMPLE rate (reference qpnet.py:263: no upsampling layer)."""
    import dataclasses
    x, h, t, d, b = synth.train_inputs(cfg_u, bl, seed, ml)
    cfg0 = dataclasses.replace(cfg_u, upsampling_factor=0)
    h0 = np.ascontiguousarray(np.repeat(h, cfg_u.upsampling_factor, axis=2)[:, :, :x.shape[1] + 3])
    return cfg0, x, h0, t, d, b


@pytest.mark.parametrize("cfgname", ["tiny", "paper"])
def test_upsampling_factor_zero_train_vs_oracle(cfgname, cuda):
    """upsampling_factor = 0 (reference src/nets/qpnet.py:203,263: no ConvTranspose2d, h is consumed as given, aligned at its END):
    logits, loss and every gradient tensor against the numpy oracle; the state_dict has no upsampling.* keys."""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd.config import TINY, PAPER
    cfg0, x, h0, t, d, b = _u0_inputs(TINY if cfgname == "tiny" else PAPER, 700, 33, 5000)
    flat = synth.make_weights(cfg0, 17)
    m = util.build_model(cfg0, flat, cuda).train()
    assert not any(k.startswith("upsampling") for k in m.state_dict())
    BL = int(b[0])
    xt, ht, tt, dt, bt = _to(cuda, x, h0, t, d, b)
    logits = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg0.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    lg, caches = TO.forward(cfg0, flat, x, h0, d, b)
    oloss, dl = TO.ce_loss(lg, t[:, -BL:])
    assert abs(loss.item() - float(oloss)) < 1e-4                  # north_star tolerance
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lg, atol=2e-5, rtol=0)
    util.assert_grads_match_oracle(TO, cfg0, flat, caches, dl, grad, a_scale=2e-5, a_rel=1e-4)


def test_full_size_batch2_step_vs_oracle(cuda):
    """Two full-size chunks per step (batch_length 20000, B = 2: the rows of both items in every contraction, the stack queue's positions
    over two batch items): loss within the north_star tolerance and every gradient tensor against the numpy oracle."""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    x, h, t, d, b = synth.train_inputs(cfg, 20000, 5017, 30000, f0_lo=55.0, f0_hi=300.0)
    xs = np.random.RandomState(9).randint(0, cfg.n_quantize, size=x.shape[1] + 1).astype(np.int64)
    x = np.stack([x[0], xs[:-1]]); t = np.stack([t[0], xs[1:]])      # second row: same features (one chunk geometry), another waveform
    h = np.concatenate([h, h]); d = np.concatenate([d, d]); b = np.concatenate([b, b])
    BL = int(b[0])
    m = util.build_model(cfg, flat, cuda).train()
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    logits = m(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    oloss, dl = TO.ce_loss(lg, t[:, -BL:])
    assert abs(loss.item() - float(oloss)) < 1e-4
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lg, atol=5e-5, rtol=0)
    util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-4, a_rel=2e-3)


def test_forward_maxd_bound_on_paper_with_a_long_chunk(cuda):
    """train.forward_maxd on the BASELINE geometry with a chunk LONGER than RF + batch_length (the bound then exceeds the true ceil(max d)
    by far: the layers' row ranges, the tap tables and the stack queue's tile counts all change): same logits as with the exact value."""
    import torch
    from qpnet_amd.config import PAPER
    from qpnet_amd.train import forward_maxd
    cfg = PAPER
    m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda)
    x, h, t, d, b = synth.train_inputs(cfg, 6000, 43, 30000)
    xt, ht, dt, _ = _to(cuda, x, h, d, b)
    b2 = torch.tensor([int(b[0]) - 2500])                      # the same chunk with a shorter batch_length
    true_maxd = int(np.ceil(d).max())
    assert forward_maxd(m, x.shape[1], h.shape[2], d.shape[1], int(b2[0]), dt) > true_maxd + 100
    with torch.no_grad():
        lg_bound = m(xt, ht, dt, b2).cpu().numpy()
        m.read_back_maxd = True
        lg_exact = m(xt, ht, dt, b2).cpu().numpy()
    np.testing.assert_allclose(lg_bound, lg_exact, atol=1e-6, rtol=0)


def test_stack_queue_equals_per_layer_launches(cuda, monkeypatch):
    """The one-launch residual stack (csrc/train_stack.hip: a work queue over (layer, tile) positions, rows handed between workgroups
    write-through behind flag words) runs the per-layer kernels' arithmetic in their order: logits BIT-identical to QPN_STACK_QUEUE=0,
    on a full-size chunk and on two batch items; its counters are readable and no wait was abandoned."""
    import ctypes as C
    import torch
    from qpnet_amd import _lib
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    for bl, seed, batch in ((20000, 5000, 1), (900, 78, 2)):
        x, h, t, d, b = synth.train_inputs(cfg, bl, seed, 30000, f0_lo=55.0, f0_hi=300.0)
        if batch == 2:
            x = np.concatenate([x, (x + 7) % cfg.n_quantize]); h = np.concatenate([h, h]); d = np.concatenate([d, d]); b = np.concatenate([b, b])
        xt, ht, dt, bt = _to(cuda, x, h, d, b)
        outs = []
        for q in ("1", "0"):
            monkeypatch.setenv("QPN_STACK_QUEUE", q)
            m = util.build_model(cfg, flat, cuda)
            with torch.no_grad():
                outs.append(m(xt, ht, dt, bt).cpu().numpy())
            if q == "1":
                st = (C.c_uint * 16)()
                _lib.check(_lib.lib().qpn_train_stack_stats(m._handle, st, 16, None))
                assert st[1] == 0                                  # the abort word
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_stack_queue_that_gives_up_is_reported_and_replaced(cuda):
    """Every wait of the one-launch residual stack is bounded.  With the published flags made unrecognisable (the situation of a peer
    workgroup that never runs) the first dependent tile's wait runs out, the launch drains, the step is REPORTED as invalid (status bit 4 ->
    QPN_ENODEV) -- and the handle runs a launch per layer from then on: the next forward is correct.  The hook exists only in the -DQPN_TESTING
    build of the library: tests/giveup_child.py `stack`, in a child process bound to that build."""
    util.run_giveup_child("stack", "QPN_TEST_STACK_GIVES_UP")


def test_deep_network_train_vs_reference(cuda, golden_dir):
    """'Rd10Rr3Ed4Er1' (reference src/utils/param_model.py:66-72: 34 residual layers -- beyond the 32 the training kernels used to carry --, fixed
    dilations up to 512, max_length 22500) at the paper-size widths: forward logits and loss, per-tensor gradients (numpy oracle) and two fused Adam
    steps against the reference's fixtures (forward_deep.npz / train_deep.npz).  The stack runs as the one-launch work queue over 34 layers, the post-net
    on the generic 16-row kernels (the skip sum is K = 34 x 64)."""
    import torch
    from cases import FORWARD_CASES_DEEP, TRAIN_CASES_DEEP
    from oracle import train_oracle as TO
    from qpnet_amd.train import FusedTrainer
    name, cfg, wseed, dseed, bl, ml = FORWARD_CASES_DEEP[0]
    g = np.load(golden_dir + "/forward_deep.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda)
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, ml)
    with torch.no_grad():
        lg = m(*_to(cuda, x, h, d, b)).cpu().numpy()
    np.testing.assert_allclose(lg, g[name + "_logits"], atol=5e-5, rtol=0)
    BL = int(b[0])
    lse = np.log(np.exp(lg[0].astype(np.float64)).sum(1))
    assert abs((lse - lg[0][np.arange(BL), t[0, -BL:]]).mean() - float(g[name + "_loss"])) < 1e-4
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES_DEEP[0]
    g = np.load(golden_dir + "/train_deep.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda).train()
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, 22500)
    xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
    BL = int(b[0])
    loss = torch.nn.CrossEntropyLoss()(m(xt, ht, dt, bt).reshape(-1, cfg.n_quantize), tt[:, -BL:].reshape(-1))
    loss.backward()
    assert abs(loss.item() - g[name + "_losses"][0]) < 1e-4
    grad = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    lgo, caches = TO.forward(cfg, flat, x, h, d, b)
    _, dl = TO.ce_loss(lgo, t[:, -BL:])
    og0 = TO.backward(cfg, flat, caches, dl)
    og = util.assert_grads_match_oracle(TO, cfg, flat, caches, dl, grad, a_scale=2e-5, a_rel=2e-4)
    ref = g[name + "_grad0_sample"] + (og - og0)[::97]
    assert np.abs(grad[::97] - ref).max() <= 1e-4 * np.abs(ref).max()
    m2 = util.build_model(cfg, flat, cuda).train()
    tr = FusedTrainer(m2, lr=1e-4)
    losses = [tr.step(*_to(cuda, *synth.train_inputs(cfg, bl, dseed + s, 22500))) for s in range(nsteps)]
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)
    util.assert_weights_after_adam(m2.flat_parameters().cpu().numpy()[::97], g[name + "_wfinal_sample"], 1e-4, nsteps)
