"""CPU: the oracle (oracle/qpnet_oracle.c) against the fixtures produced by importing the
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from cases import DECODE_CASES, DECODE_CASES2, DECODE_CASES_D, FORWARD_CASES, TRAIN_CASES, decode2_inputs
from qpnet_amd import synth
import util


def test_mulaw_kat(oracle, golden_dir):
    g = np.load(golden_dir + "/kat.npz")
    np.testing.assert_array_equal(oracle.encode_mu_law(g["mulaw_x"]), g["mulaw_enc"])
    np.testing.assert_allclose(oracle.decode_mu_law(np.arange(256)), g["mulaw_dec"], rtol=1e-14, atol=0)
    # SURVEY §8a a1/a2 known answers
    np.testing.assert_array_equal(oracle.encode_mu_law(np.array([-1, -.5, 0, .5, 1.0])), [0, 16, 128, 239, 255])
    np.testing.assert_allclose(oracle.decode_mu_law(np.array([0, 128, 255])), [-1.02207017, 0.0, 0.97840458], atol=1e-8)


def test_mulaw_product_matches_reference(golden_dir):
    from qpnet_amd.qpnet import encode_mu_law, decode_mu_law
    g = np.load(golden_dir + "/kat.npz")
    np.testing.assert_array_equal(encode_mu_law(g["mulaw_x"], 256), g["mulaw_enc"])
    np.testing.assert_array_equal(decode_mu_law(np.arange(256), 256), g["mulaw_dec"])


@pytest.mark.parametrize("tag", ["h", "1", "x"])
def test_dilated_index_kat(oracle, golden_dir, tag):
    g = np.load(golden_dir + "/kat.npz")
    d64 = g["didx_d64_" + tag]
    d32 = d64.astype(np.float32)
    for k in range(4):
        np.testing.assert_array_equal(oracle.dilated_index_train(d32, 2 ** k), g["didx_train_f32_%s_%d" % (tag, k)])
        np.testing.assert_array_equal(oracle.dilated_index_train(d64, 2 ** k), g["didx_train_f64_%s_%d" % (tag, k)])
        np.testing.assert_array_equal(oracle.dilated_index_gen(d32, 2 ** k), g["didx_gen_f32_%s_%d" % (tag, k)])
        np.testing.assert_array_equal(oracle.dilated_index_gen(d64, 2 ** k), g["didx_gen_f64_%s_%d" % (tag, k)])


def test_dilated_index_long(oracle, golden_dir):
    g = np.load(golden_dir + "/kat.npz")
    np.testing.assert_array_equal(oracle.dilated_index_train(g["didx_long_d32"], 8), g["didx_long_train_f32_3"])


@pytest.mark.parametrize("case", DECODE_CASES, ids=[c[0] for c in DECODE_CASES])
def test_oracle_decode_equals_reference_streams(case, oracle, golden_dir):
    """bit-exact mu-law indices for greedy decode, incl. B>1 completion order, f0 x0.5 / x1.5,
    float32 (extra_memory) and float64 index paths."""
    name, cfg, wseed, utts, extra = case
    g = np.load(golden_dir + "/decode.npz")
    flat = synth.make_weights(cfg, wseed)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    outs = oracle.batch_fast_generate(cfg, flat, bx, bh, nlist, bd.astype(np.float32) if extra else bd)
    assert nlist == list(g[name + "_nleft"])
    for i, s in enumerate(outs):
        np.testing.assert_array_equal(s, g["%s_out%d" % (name, i)].astype(np.int64))


@pytest.mark.parametrize("case", DECODE_CASES_D, ids=[c[0] for c in DECODE_CASES_D])
def test_oracle_default_geometry_decode_equals_reference_streams(case, oracle, golden_dir):
    """the repo-default geometry (C=512, 12 fixed + 4 adaptive layers: what runQP.py builds) pinned directly to the
    reference's greedy streams: 2 199 samples at B=1 and a B=2 batch of unequal lengths at F0 x 1.5 (decode_d.npz)."""
    name, cfg, wseed, utts, extra = case
    g = np.load(golden_dir + "/decode_d.npz")
    flat = synth.make_weights(cfg, wseed)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    outs = oracle.batch_fast_generate(cfg, flat, bx, bh, nlist, bd)
    assert nlist == list(g[name + "_nleft"])
    for i, s in enumerate(outs):
        np.testing.assert_array_equal(s, g["%s_out%d" % (name, i)].astype(np.int64))


@pytest.mark.parametrize("case", FORWARD_CASES, ids=[c[0] for c in FORWARD_CASES])
def test_oracle_forward_equals_reference_logits(case, oracle, golden_dir):
    """QPNet.forward (teacher forced) == streaming oracle on the same samples (SURVEY §4 identity),
    logits within fp32 reassociation error and CE loss within 1e-4 (north_star tolerance)."""
    name, cfg, wseed, dseed, bl, ml = case
    g = np.load(golden_dir + "/forward.npz")
    flat = synth.make_weights(cfg, wseed)
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, ml)
    BL = int(b[0])
    assert BL == int(g[name + "_bl"])
    lg = oracle.forward(cfg, flat, x[0], h[0], d[0], BL)
    ref = g[name + "_logits"][0]
    np.testing.assert_allclose(lg, ref, atol=5e-6, rtol=0)
    assert (lg.argmax(1) == ref.argmax(1)).all()
    lse = np.log(np.exp(lg.astype(np.float64)).sum(1))
    loss = (lse - lg[np.arange(BL), t[0, -BL:]]).mean()
    assert abs(loss - float(g[name + "_loss"])) < 1e-4


def test_qexp_qgate_accuracy(oracle):
    """the spec's exp / gate are within a few ulp of libm (so the fixed-order spec is a faithful fp32
    evaluation of sigmoid*tanh, reference qpnet.py:634-635)."""
    L = oracle.lib()
    xs = np.linspace(-20, 20, 4001).astype(np.float32)
    e = np.array([L.qpo_qexp(float(x)) for x in xs], dtype=np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 3e-7
    rs = np.random.RandomState(0)
    a, b = rs.uniform(-8, 8, 2000).astype(np.float32), rs.uniform(-8, 8, 2000).astype(np.float32)
    gq = np.array([L.qpo_qgate(float(u), float(v)) for u, v in zip(a, b)])
    gr = 1 / (1 + np.exp(-a.astype(np.float64))) * np.tanh(b.astype(np.float64))
    assert np.max(np.abs(gq - gr)) < 3e-7


def test_philox_known_answer(oracle):
    """Random123 known-answer vector: Philox4x32-10, counter 0, key 0 -> 0x6627e8d5 (first word)."""
    import ctypes as C
    L = oracle.lib()
    L.qpo_philox_first.restype = C.c_uint32
    L.qpo_philox_first.argtypes = [C.c_uint32] * 4
    assert L.qpo_philox_first(0, 0, 0, 0) == 0x6627E8D5


def test_sampling_spec_is_softmax_distributed(oracle):
    """statistical parity with the reference's softmax + Categorical (qpnet.py:507-510): the empirical law of the
    spec sampler on teacher-forced logits matches softmax(logits) (chi-square over the classes with mass)."""
    from qpnet_amd.config import TINY
    cfg = TINY
    flat = synth.make_weights(cfg, 11, gain=3.0)
    x, h, d, n = synth.decode_inputs(cfg, 3, 5, 1.0)
    teacher = np.full(n, 128, dtype=np.int64)      # constant input -> (near) stationary logits after the warm-up
    r = oracle.decode(cfg, flat, h, d, x, n, teacher=teacher, want_logits=True, mode="sampling", seed=99)
    lg = r["logits"][-1].astype(np.float64)
    p = np.exp(lg - lg.max()); p /= p.sum()
    # draw many samples from the SAME logits by varying the seed (row/step counters change the stream)
    draws = []
    for s in range(40):
        rr = oracle.decode(cfg, flat, h, d, x, n, teacher=teacher, mode="sampling", seed=1000 + s)
        draws.append(rr["samples"][-60:])
    draws = np.concatenate(draws)
    # logits over the last 60 steps are identical up to the aux frame; compare against their mean law
    lgs = r["logits"][-60:].astype(np.float64)
    pm = np.exp(lgs - lgs.max(1, keepdims=True)); pm /= pm.sum(1, keepdims=True); pm = pm.mean(0)
    cnt = np.bincount(draws, minlength=256).astype(np.float64)
    mask = pm * draws.size >= 5
    chi2 = ((cnt[mask] - pm[mask] * draws.size) ** 2 / (pm[mask] * draws.size)).sum()
    dof = mask.sum() - 1
    assert chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)


@pytest.mark.parametrize("case", DECODE_CASES2, ids=[c["name"] for c in DECODE_CASES2])
def test_oracle_decode_worst_case_pitch_and_long_seeds(case, oracle, golden_dir):
    """the oracle against reference streams at maxd ~ 123 (45 Hz x 0.5, > 5 k samples) and with n_x > 1 seeds"""
    cfg, name, extra = case["cfg"], case["name"], case["extra"]
    g = np.load(golden_dir + "/decode2.npz")
    flat = synth.make_weights(cfg, case["wseed"])
    bx, bh, bd, ns = decode2_inputs(case)
    nlist = list(ns)
    outs = oracle.batch_fast_generate(cfg, flat, bx, bh, nlist, bd.astype(np.float32) if extra else bd)
    assert nlist == list(g[name + "_nleft"])
    for i, s in enumerate(outs):
        np.testing.assert_array_equal(s, g["%s_out%d" % (name, i)].astype(np.int64))


def test_train_oracle_pinned_to_reference_autograd(golden_dir):
    """oracle/train_oracle.py (numpy forward + hand-derived backward + Adam) against the fixture the imported reference
    produced with torch autograd + torch.optim.Adam: loss of every step, the FULL gradient of step 0, final weights."""
    from oracle import train_oracle as TO
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES[0]          # tiny: the fixture holds the whole gradient
    g = np.load(golden_dir + "/train.npz")
    flat = synth.make_weights(cfg, wseed)
    opt = TO.Adam(flat.size)
    losses = []
    for step in range(nsteps):
        x, h, t, d, b = synth.train_inputs(cfg, bl, dseed + step, 30000)
        loss, grad = TO.train_step(cfg, flat, opt, x, h, t, d, b)
        losses.append(loss)
        if step == 0:
            ref = g[name + "_grad0"]
            assert np.abs(grad - ref).max() <= 2e-5 * np.abs(ref).max()
            offs, _ = cfg.param_offsets()
            norms = g[name + "_grad0_norms"]
            for (k, (o, shp)), nr in zip(offs.items(), norms):
                n = int(np.prod(shp))
                assert abs(np.linalg.norm(grad[o:o + n]) - nr) <= 1e-4 * max(nr, 1e-6) + 1e-7, k
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)        # north_star tolerance
    np.testing.assert_allclose(flat[::97], g[name + "_wfinal_sample"], atol=2e-6, rtol=0)


def test_train_torch_oracle_pinned_to_reference_autograd(golden_dir):
    """oracle/train_torch.py (the torch-CPU restatement bench.py times as `cpu_baseline`) against the same reference fixture: loss of every
    step, the full gradient of step 0, final weights after the Adam steps."""
    from oracle import train_torch as TT
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES[0]
    g = np.load(golden_dir + "/train.npz")
    tr = TT.Trainer(cfg, synth.make_weights(cfg, wseed))
    losses = []
    for step in range(nsteps):
        x, h, t, d, b = synth.train_inputs(cfg, bl, dseed + step, 30000)
        if step == 0:
            loss, grad = tr.loss_and_grad(x, h, t, d, b)
            ref = g[name + "_grad0"]
            assert np.abs(grad - ref).max() <= 2e-5 * np.abs(ref).max()
            tr.opt.step()
        else:
            loss = tr.step(x, h, t, d, b)
        losses.append(loss)
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(tr.flat.detach().numpy()[::97], g[name + "_wfinal_sample"], atol=2e-6, rtol=0)


def test_train_oracle_paper_loss_and_grad_sample(golden_dir):
    """paper-size: first-step loss and the strided gradient sample of the reference fixture"""
    from oracle import train_oracle as TO
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES[1]
    g = np.load(golden_dir + "/train.npz")
    flat = synth.make_weights(cfg, wseed)
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, 30000)
    lg, caches = TO.forward(cfg, flat, x, h, d, b)
    BL = int(b[0])
    loss, dl = TO.ce_loss(lg, t[:, -BL:])
    assert abs(loss - g[name + "_losses"][0]) < 1e-4
    grad = TO.backward(cfg, flat, caches, dl)
    ref = g[name + "_grad0_sample"]
    assert np.abs(grad[::97] - ref).max() <= 1e-4 * np.abs(ref).max()


def test_train_oracle_default_geometry(golden_dir):
    """the repo-default geometry (C=512, 12 fixed + 4 adaptive layers): numpy oracle vs the reference's logits / loss /
    gradient sample / two Adam steps (fixtures forward_d.npz, train_d.npz; short chunks)"""
    from cases import FORWARD_CASES_D, TRAIN_CASES_D
    from oracle import train_oracle as TO
    name, cfg, wseed, dseed, bl, ml = FORWARD_CASES_D[0]
    g = np.load(golden_dir + "/forward_d.npz")
    flat = synth.make_weights(cfg, wseed)
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, ml)
    lg, _ = TO.forward(cfg, flat, x, h, d, b)
    np.testing.assert_allclose(lg, g[name + "_logits"], atol=5e-5, rtol=0)
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES_D[0]
    g = np.load(golden_dir + "/train_d.npz")
    flat = synth.make_weights(cfg, wseed)
    opt = TO.Adam(flat.size)
    losses = []
    for step in range(nsteps):
        x, h, t, d, b = synth.train_inputs(cfg, bl, dseed + step, 2000)
        loss, grad = TO.train_step(cfg, flat, opt, x, h, t, d, b)
        losses.append(loss)
        if step == 0:
            ref = g[name + "_grad0_sample"]
            assert np.abs(grad[::97] - ref).max() <= 1e-4 * np.abs(ref).max()
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(flat[::97], g[name + "_wfinal_sample"], atol=2e-6, rtol=0)


# ---------------------------------------------------------------- round 5: the reference's deep network 'Rd10Rr3Ed4Er1' (34 layers, fixed dilations up to 512)
def test_oracle_deep_network_decode_equals_reference_stream(oracle, golden_dir):
    """src/utils/param_model.py:66-72 at the paper-size widths: the C oracle against the reference's greedy stream (decode_deep.npz: 2 199 samples; the
    fixed stack's rings are 2 x 512 rows deep, the receptive field 3 069 + 15 maxd + 1 samples)."""
    from cases import DECODE_CASES_DEEP
    name, cfg, wseed, utts, extra = DECODE_CASES_DEEP[0]
    g = np.load(golden_dir + "/decode_deep.npz")
    flat = synth.make_weights(cfg, wseed)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    outs = oracle.batch_fast_generate(cfg, flat, bx, bh, nlist, bd)
    assert nlist == list(g[name + "_nleft"])
    for i, s in enumerate(outs):
        np.testing.assert_array_equal(s, g["%s_out%d" % (name, i)].astype(np.int64))


def test_train_oracle_deep_network(golden_dir):
    """the same network: numpy oracle vs the reference's logits / loss / gradient sample / two Adam steps (forward_deep.npz, train_deep.npz)"""
    from cases import FORWARD_CASES_DEEP, TRAIN_CASES_DEEP
    from oracle import train_oracle as TO
    name, cfg, wseed, dseed, bl, ml = FORWARD_CASES_DEEP[0]
    g = np.load(golden_dir + "/forward_deep.npz")
    flat = synth.make_weights(cfg, wseed)
    x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, ml)
    lg, _ = TO.forward(cfg, flat, x, h, d, b)
    np.testing.assert_allclose(lg, g[name + "_logits"], atol=5e-5, rtol=0)
    name, cfg, wseed, dseed, bl, nsteps = TRAIN_CASES_DEEP[0]
    g = np.load(golden_dir + "/train_deep.npz")
    flat = synth.make_weights(cfg, wseed)
    opt = TO.Adam(flat.size)
    losses, grads = [], []
    for step in range(nsteps):
        x, h, t, d, b = synth.train_inputs(cfg, bl, dseed + step, 22500)
        loss, grad = TO.train_step(cfg, flat, opt, x, h, t, d, b)
        losses.append(loss); grads.append(grad)
        if step == 0:
            ref = g[name + "_grad0_sample"]
            assert np.abs(grad[::97] - ref).max() <= 1e-4 * np.abs(ref).max()
    np.testing.assert_allclose(losses, g[name + "_losses"], atol=1e-4, rtol=0)
    util.assert_weights_after_adam(flat[::97], g[name + "_wfinal_sample"], 1e-4, nsteps, significant=util.significant_elements(cfg, grads)[::97])
