"""GPU parity: the HIP persistent decode kernel vs (a) the reference's own greedy streams
(golden fixtures made by importing the reference) and (b) the CPU oracle, bit-exact."""
import numpy as np
import pytest

from cases import DECODE_CASES, DECODE_CASES2, DECODE_CASES_D, decode2_inputs
from qpnet_amd import synth
import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", DECODE_CASES, ids=[c[0] for c in DECODE_CASES])
def test_decode_matches_reference_streams(case, cuda, golden_dir, oracle, monkeypatch):
    """one workgroup (one CU) per utterance: k_decode_fast / k_decode (the pipelined default of the paper-size geometry has
    its own tests below)"""
    import torch
    monkeypatch.setenv("QPN_DECODE_PIPE", "0")
    name, cfg, wseed, utts, extra = case
    g = np.load(golden_dir + "/decode.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    d_arg = torch.from_numpy(bd).float().to(cuda) if extra else bd
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, d_arg,
                                 mode="argmax", extra_memory=extra)
    assert nlist == list(g[name + "_nleft"])           # list consumed like the reference
    onl = list(ns)
    o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, onl, bd.astype(np.float32) if extra else bd)
    assert len(outs) == len(utts)
    for i, s in enumerate(outs):
        ref = g["%s_out%d" % (name, i)].astype(np.int64)
        assert s.dtype == np.int64 and s.shape == ref.shape
        np.testing.assert_array_equal(s, o_outs[i], err_msg="HIP vs oracle, row %d" % i)
        np.testing.assert_array_equal(s, ref, err_msg="HIP vs reference stream, row %d" % i)


def test_stream_logits_bitwise_vs_oracle(cuda, oracle):
    """Teacher-forced per-step logits of the HIP kernel are BIT-identical to the oracle's."""
    import torch
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 77)
    m = util.build_model(cfg, flat, cuda)
    x, h, d, n = synth.decode_inputs(cfg, 6, 5, 1.0)
    rs = np.random.RandomState(9)
    teacher = rs.randint(0, 256, size=n).astype(np.int64)
    out, logits = m._stream_logits(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda),
                                   d[None], torch.from_numpy(teacher[None]), n)
    r = oracle.decode(cfg, flat, h, d, x, n, teacher=teacher, want_logits=True)
    lg = logits[0].cpu().numpy()
    assert np.array_equal(lg.view(np.uint32), r["logits"].view(np.uint32)), \
        "max abs diff %g" % np.abs(lg - r["logits"]).max()
    np.testing.assert_array_equal(out[0].cpu().numpy(), r["samples"])


def test_decode_range_error(cuda):
    """A dilated factor above maxd must be reported (reference asserts, qpnet.py:294,417)."""
    import torch
    from qpnet_amd.config import TINY
    from qpnet_amd import _lib
    flat = synth.make_weights(TINY, 3)
    m = util.build_model(TINY, flat, cuda)
    x, h, d, n = synth.decode_inputs(TINY, 4, 5, 1.0)
    d = d.copy(); d[200:] = 0.2        # rounds to a tap distance of 0 -> out of contract
    with pytest.raises(_lib.QpnError) as e:
        m.batch_fast_generate(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), [n], d[None], mode="argmax")
    assert e.value.code == -4


@pytest.mark.parametrize("cfgname", ["tiny", "paper"])
def test_sampling_mode_bitwise_vs_oracle(cfgname, cuda, oracle):
    """mode="sampling" (what the reference decode script runs by default, qpnet_decode.py:312-314): softmax +
    inverse-CDF draw with Philox4x32-10 in the spec order -> every draw equals the oracle's, B=2 unequal lengths."""
    import torch
    from qpnet_amd.config import TINY, PAPER
    cfg = TINY if cfgname == "tiny" else PAPER
    flat = synth.make_weights(cfg, 31)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, [(61, 9, 1.0), (62, 6, 1.0)])
    m.sampling_seed = 0x1234567890ABCDEF
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="sampling")
    o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd, mode="sampling", seed=0x1234567890ABCDEF)
    for a, b in zip(outs, o_outs):
        np.testing.assert_array_equal(a, b)
    # different seed -> different stream; sampling is not argmax
    m.sampling_seed = 7
    outs2 = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="sampling")
    assert not np.array_equal(outs[0], outs2[0])
    assert len(np.unique(outs[1])) > 32


def test_full_size_decode_properties(cuda, oracle):
    """BASELINE config[3]/[4] size (10 s @22.05 kHz = 2005 frames -> 220 549 samples, F0 x1.0 / x0.5 / x1.5 in one batch):
    compared with the oracle over the full length (3 x 220 549 samples, bit-exact) and through size-independent properties --
      prefix:   an autoregressive stream does not depend on how long the utterance goes on, so the first samples of the 10 s
                streams equal a 40-frame decode of the same features, which IS compared with the oracle bit by bit;
      batch:    a row's stream does not depend on its batch mates (alone == in the batch of three);
      repeat:   two launches give identical streams; ordering = ascending length, ties in input order;
      kernels:  three independently written kernels (pipelined five-CU, one-CU, cooperative row-sliced) agree on every sample."""
    import torch
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 31)
    m = util.build_model(cfg, flat, cuda)
    F, Fs = 2005, 40
    U = cfg.upsampling_factor
    utts = [(700, F, 1.0), (701, F, 0.5), (702, F, 1.5)]
    x, h, d, ns = synth.decode_batch(cfg, utts)
    xt, ht = torch.from_numpy(x).to(cuda), torch.from_numpy(h).to(cuda)
    ys = m.batch_fast_generate(xt, ht, list(ns), d, mode="argmax")
    assert [len(y) for y in ys] == [F * U - 1] * 3
    ys2 = m.batch_fast_generate(xt, ht, list(ns), d, mode="argmax")
    for a, b in zip(ys, ys2):
        np.testing.assert_array_equal(a, b)
    maxd = int(np.nanmax(np.ceil(d)))                       # the batch's receptive field (reference qpnet.py:347-350)
    for i, (seed, _, fac) in enumerate(utts):
        assert ys[i].min() >= 0 and ys[i].max() < cfg.n_quantize
        # prefix property: the first 40 frames' worth of every 10 s stream == the oracle run on the 40-frame prefix with the
        # same receptive field (maxd decides the warm-up padding, so it must be the batch's)
        xs, hs, ds, _ = synth.decode_inputs(cfg, F, seed, fac)
        hs, ds = np.ascontiguousarray(hs[:, :Fs]), ds[:Fs * U]
        ref = oracle.decode(cfg, flat, hs, ds, xs, Fs * U - 1, maxd=maxd)["samples"]
        np.testing.assert_array_equal(ys[i][:Fs * U - 1], ref)
    # batch independence: the x0.5 row alone
    assert int(np.ceil(d[1]).max()) == maxd                 # the halved-F0 row sets the batch's maxd, so alone == in the batch
    alone = m.batch_fast_generate(xt[1:2], ht[1:2], [ns[1]], d[1:2], mode="argmax")[0]
    np.testing.assert_array_equal(alone, ys[1])
    # independent implementations: the streams above come from the pipelined five-CU kernel (decode_pipe.hip); the one-CU
    # kernel (decode.hip: different tiling, synchronisation and data flow, same arithmetic spec) must give the same 3 x 220 549
    # samples, and so must the cooperative row-sliced kernel (decode_coop.hip) on the worst-pitch row
    # (the launch-plan knobs are read when a module's native handle is created: a module per kernel)
    import os
    try:
        os.environ["QPN_DECODE_PIPE"] = "0"
        ys3 = util.build_model(cfg, flat, cuda).batch_fast_generate(xt, ht, list(ns), d, mode="argmax")
        os.environ["QPN_DECODE_COOP"] = "4"
        coop = util.build_model(cfg, flat, cuda).batch_fast_generate(xt[1:2], ht[1:2], [ns[1]], d[1:2], mode="argmax")[0]
    finally:
        os.environ.pop("QPN_DECODE_PIPE", None); os.environ.pop("QPN_DECODE_COOP", None)
    for a, b in zip(ys, ys3):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(coop, ys[1])
    # and the oracle itself over the FULL 10 s of all three rows (the C port does ~40 k samples/s per core: one thread per row)
    from concurrent.futures import ThreadPoolExecutor

    def full(i):
        seed, _, fac = utts[i]
        xs, hs, ds, n = synth.decode_inputs(cfg, F, seed, fac)
        return oracle.decode(cfg, flat, hs, ds, xs, n, maxd=maxd)["samples"]
    with ThreadPoolExecutor(max_workers=3) as ex:
        refs = list(ex.map(full, range(3)))
    for i in range(3):
        np.testing.assert_array_equal(ys[i], refs[i], err_msg="full 10 s stream of row %d vs the oracle" % i)


def test_headline_decode_batch_rows_full_length_vs_oracle(cuda, oracle):
    """the batch bench.py's `decode` object times -- BASELINE config[3], 20 utterances x 2005 frames (220 549 samples each, the reference's
    decode_batch_size, runQP.py:66; the same feature seeds 100..119) through the five-role pipelined kernel, 20 groups resident at once -- with rows 0, 7, 13 and
    19 compared with the C oracle over their FULL length, bit for bit (reference src/nets/qpnet.py:446-557), every row's range and length checked, and
    the launch repeated once (identical streams)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    m = util.build_model(cfg, flat, cuda)
    F, B = 2005, 20
    utts = [(100 + b, F, 1.0) for b in range(B)]
    x, h, d, ns = synth.decode_batch(cfg, utts)
    xt, ht = torch.from_numpy(x).to(cuda), torch.from_numpy(h).to(cuda)
    ys = m.batch_fast_generate(xt, ht, list(ns), d, mode="argmax")
    assert "pipe" in getattr(m, "last_decode_plan", "pipe")
    assert [len(y) for y in ys] == [F * cfg.upsampling_factor - 1] * B
    assert all(y.min() >= 0 and y.max() < cfg.n_quantize for y in ys)
    maxd = int(np.nanmax(np.ceil(d)))                       # the batch's receptive field (reference qpnet.py:347-350)
    rows = (0, 7, 13, 19)

    def full(i):
        xs, hs, ds, n = synth.decode_inputs(cfg, F, utts[i][0], 1.0)
        return oracle.decode(cfg, flat, hs, ds, xs, n, maxd=maxd)["samples"]
    with ThreadPoolExecutor(max_workers=4) as ex:
        refs = list(ex.map(full, rows))
    for i, ref in zip(rows, refs):                          # (equal lengths: completion order = input order)
        np.testing.assert_array_equal(ys[i], ref, err_msg="full 10 s stream of row %d of the 20-row batch vs the oracle" % i)
    ys2 = m.batch_fast_generate(xt, ht, list(ns), d, mode="argmax")
    for a, b2 in zip(ys, ys2):
        np.testing.assert_array_equal(a, b2)


@pytest.mark.parametrize("geo", [(128, 128, 2, 1, 2, 1), (96, 256, 2, 1, 1, 1), (64, 128, 3, 2, 2, 1)], ids=["C128", "C96", "F3x2"])
def test_other_geometries_bitwise_vs_oracle(geo, cuda, oracle):
    """Depth/repeat/width combinations outside the BASELINE configs (the repo default uses repeat 3): interpreter kernel for
    n_resch 96 / 128, specialised kernel with another skip width and a repeated fixed stack for n_resch 64."""
    import torch
    from qpnet_amd.config import QPNetConfig
    C, S, fd, fr, ad, ar = geo
    cfg = QPNetConfig(n_resch=C, n_skipch=S, dilationF_depth=fd, dilationF_repeat=fr, dilationA_depth=ad, dilationA_repeat=ar)
    flat = synth.make_weights(cfg, 7)
    m = util.build_model(cfg, flat, cuda)
    x, h, d, n = synth.decode_inputs(cfg, 5, 11, 1.5)
    y = m.batch_fast_generate(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), [n], d[None], mode="argmax")[0]
    np.testing.assert_array_equal(y, oracle.decode(cfg, flat, h, d, x, n)["samples"])


@pytest.mark.parametrize("cfgname", ["tiny", "paper"])
def test_upsampling_factor_zero_decode_vs_oracle(cfgname, cuda, oracle):
    """upsampling_factor = 0 (reference src/nets/qpnet.py:203,343: no upsampling layer, the features arrive at sample rate): greedy and
    sampling streams against the C oracle; n_samples counts samples directly."""
    import dataclasses
    import torch
    from qpnet_amd.config import TINY, PAPER
    cfg_u = TINY if cfgname == "tiny" else PAPER
    cfg0 = dataclasses.replace(cfg_u, upsampling_factor=0)
    flat = synth.make_weights(cfg0, 19)
    m = util.build_model(cfg0, flat, cuda)
    x, h, d, n = synth.decode_inputs(cfg_u, 7, 23, 1.0)
    h0 = np.ascontiguousarray(np.repeat(h, cfg_u.upsampling_factor, axis=1))          # (n_aux, T): one feature column per sample
    assert h0.shape[1] == d.size == n + 1
    for mode in ("argmax", "sampling"):
        m.sampling_seed = 99
        y = m.batch_fast_generate(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h0[None]).to(cuda), [n], d[None], mode=mode)[0]
        r = oracle.decode(cfg0, flat, h0, d, x, n, mode=mode, seed=99, row=0)
        np.testing.assert_array_equal(y, r["samples"], err_msg=mode)


def test_other_class_count_decode_and_train_vs_oracle(cuda, oracle):
    """n_quantize = 128 (the reference takes it as a constructor argument, qpnet.py:174): greedy and sampling streams bit-exact against
    the C oracle (the sampling spec lays Q / 64 classes on a lane), training logits / loss / gradients against the numpy oracle; a class
    count the sampling spec does not cover is refused with a message, not mis-sampled."""
    import torch
    from oracle import train_oracle as TO
    from qpnet_amd import _lib
    from qpnet_amd.config import QPNetConfig
    cfg = QPNetConfig(n_quantize=128, n_resch=64, n_skipch=128, dilationF_depth=2, dilationF_repeat=1, dilationA_depth=2, dilationA_repeat=1)
    flat = synth.make_weights(cfg, 29)
    m = util.build_model(cfg, flat, cuda)
    x, h, d, n = synth.decode_inputs(cfg, 6, 31, 1.0)
    for mode in ("argmax", "sampling"):
        m.sampling_seed = 5
        y = m.batch_fast_generate(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), [n], d[None], mode=mode)[0]
        np.testing.assert_array_equal(y, oracle.decode(cfg, flat, h, d, x, n, mode=mode, seed=5, row=0)["samples"], err_msg=mode)
        assert y.max() < 128
    xb, hb, tb, db, bb = synth.train_inputs(cfg, 500, 8, 4000)
    BL = int(bb[0])
    mt = util.build_model(cfg, flat, cuda).train()
    xt, ht, tt, dt, bt = [torch.from_numpy(np.ascontiguousarray(a)).to(cuda) for a in (xb, hb, tb, db, bb)]
    logits = mt(xt, ht, dt, bt)
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, 128), tt[:, -BL:].reshape(-1))
    loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in mt.parameters()]).cpu().numpy()
    lg, caches = TO.forward(cfg, flat, xb, hb, db, bb)
    oloss, dl = TO.ce_loss(lg, tb[:, -BL:])
    og = TO.backward(cfg, flat, caches, dl)
    assert abs(loss.item() - float(oloss)) < 1e-4
    np.testing.assert_allclose(logits.detach().cpu().numpy(), lg, atol=2e-5, rtol=0)
    assert np.abs(grad - og).max() <= 2e-5 * np.abs(og).max()
    cfg96 = QPNetConfig(n_quantize=96, n_resch=32, n_skipch=32, dilationF_depth=2, dilationF_repeat=1, dilationA_depth=1, dilationA_repeat=1)
    m96 = util.build_model(cfg96, synth.make_weights(cfg96, 3), cuda)
    x, h, d, n = synth.decode_inputs(cfg96, 3, 31, 1.0)
    with pytest.raises(_lib.QpnError) as e:
        m96.batch_fast_generate(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), [n], d[None], mode="sampling")
    assert "n_quantize" in str(e.value)


@pytest.mark.parametrize("case", DECODE_CASES2, ids=[c["name"] for c in DECODE_CASES2])
def test_decode_worst_case_pitch_and_long_seeds(case, cuda, golden_dir, oracle):
    """Reference streams at the corpus pitch floor with 0.5x F0 scaling (maxd ~ 123: the deepest rings the path meets,
    > 5 k samples) and with seeds of several samples per row (n_x > 1): bit-exact vs the reference and the oracle."""
    import torch
    cfg, name, extra = case["cfg"], case["name"], case["extra"]
    g = np.load(golden_dir + "/decode2.npz")
    flat = synth.make_weights(cfg, case["wseed"])
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = decode2_inputs(case)
    assert int(np.nanmax(np.ceil(bd))) == int(g[name + "_maxd"])
    nlist = list(ns)
    d_arg = torch.from_numpy(bd).float().to(cuda) if extra else bd
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, d_arg, mode="argmax", extra_memory=extra)
    assert nlist == list(g[name + "_nleft"])
    o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd.astype(np.float32) if extra else bd)
    for i, s in enumerate(outs):
        ref = g["%s_out%d" % (name, i)].astype(np.int64)
        np.testing.assert_array_equal(s, o_outs[i], err_msg="HIP vs oracle, row %d" % i)
        np.testing.assert_array_equal(s, ref, err_msg="HIP vs reference stream, row %d" % i)


def test_decode_more_rows_than_cus(cuda, oracle):
    """B = 300 utterances in one call (more workgroups than the chip has CUs: the tail waits for a free CU):
    unequal lengths, every row equals its single-row oracle stream, completion order kept."""
    import torch
    from qpnet_amd.config import TINY
    cfg = TINY
    flat = synth.make_weights(cfg, 5)
    m = util.build_model(cfg, flat, cuda)
    utts = [(300 + (b % 7), 3 + (b % 3), 1.0) for b in range(300)]
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="argmax")
    assert [len(o) for o in outs] == sorted(ns)
    order = sorted(range(300), key=lambda i: ns[i])
    maxd = int(np.nanmax(np.ceil(bd)))
    cache = {}
    for pos in (0, 1, 57, 150, 151, 298, 299):
        i = order[pos]
        key = utts[i]
        if key not in cache:
            cache[key] = oracle.decode(cfg, flat, bh[i], bd[i], bx[i], ns[i], maxd=maxd)["samples"]
        np.testing.assert_array_equal(outs[pos], cache[key])
    # rows with identical inputs give identical streams wherever they ran
    for pos in range(300):
        i = order[pos]
        j = order.index(next(k for k in order if utts[k] == utts[i]))
        np.testing.assert_array_equal(outs[pos], outs[j])


@pytest.mark.parametrize("tag", ["h", "1", "x"])
def test_dilated_index_exports_vs_reference_kat(tag, cuda, golden_dir):
    """qpn_dilated_index_{train,gen_f32,gen_f64} (the exported _dilated_index / _generate_dilated_index) on the GPU
    against the reference's own index tensors, incl. the |idx| > 16384 float32 rounding case."""
    import torch
    from qpnet_amd import _lib
    L = _lib.lib()
    g = np.load(golden_dir + "/kat.npz")
    d64 = g["didx_d64_" + tag]
    d32 = d64.astype(np.float32)
    B, n = d32.shape
    t32 = torch.from_numpy(d32).to(cuda); t64 = torch.from_numpy(d64).to(cuda)
    for k in range(4):
        o = torch.empty((B, n), dtype=torch.int64, device=cuda)
        _lib.check(L.qpn_dilated_index_train(t32.data_ptr(), B, n, 2 ** k, o.data_ptr(), None))
        np.testing.assert_array_equal(o.cpu().numpy(), g["didx_train_f32_%s_%d" % (tag, k)])
        _lib.check(L.qpn_dilated_index_gen_f32(t32.data_ptr(), B * n, 2 ** k, o.data_ptr(), None))
        np.testing.assert_array_equal(o.cpu().numpy(), g["didx_gen_f32_%s_%d" % (tag, k)])
        o32 = torch.empty((B, n), dtype=torch.int32, device=cuda)
        _lib.check(L.qpn_dilated_index_gen_f64(t64.data_ptr(), B * n, 2 ** k, o32.data_ptr(), None))
        np.testing.assert_array_equal(o32.cpu().numpy(), g["didx_gen_f64_%s_%d" % (tag, k)])
    if tag == "h":
        dl = torch.from_numpy(g["didx_long_d32"]).to(cuda)
        o = torch.empty(dl.shape, dtype=torch.int64, device=cuda)
        _lib.check(L.qpn_dilated_index_train(dl.data_ptr(), dl.shape[0], dl.shape[1], 8, o.data_ptr(), None))
        np.testing.assert_array_equal(o.cpu().numpy(), g["didx_long_train_f32_3"])


# ---------------------------------------------------------------- several cooperating workgroups per utterance (decode_coop.hip)
@pytest.mark.parametrize("case", DECODE_CASES, ids=[c[0] for c in DECODE_CASES])
def test_cooperative_decode_matches_reference_streams(case, cuda, golden_dir, monkeypatch):
    """QPN_DECODE_COOP=4 runs the multi-workgroup kernel (row-sliced matrices, tagged-granule all-gathers) on the small
    geometries: the same reference streams, bit for bit (paper-size: 4 workgroups per utterance; tiny: 1)."""
    import torch
    monkeypatch.setenv("QPN_DECODE_COOP", "4")
    name, cfg, wseed, utts, extra = case
    g = np.load(golden_dir + "/decode.npz")
    m = util.build_model(cfg, synth.make_weights(cfg, wseed), cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    d_arg = torch.from_numpy(bd).float().to(cuda) if extra else bd
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, d_arg, mode="argmax", extra_memory=extra)
    assert nlist == list(g[name + "_nleft"])
    for i, s in enumerate(outs):
        np.testing.assert_array_equal(s, g["%s_out%d" % (name, i)].astype(np.int64), err_msg="cooperative HIP vs reference stream, row %d" % i)


def test_cooperative_decode_sampling_and_logits(cuda, oracle, monkeypatch):
    """sampling mode and teacher-forced logits through the cooperative kernel: bit-identical to the oracle"""
    import torch
    from qpnet_amd.config import PAPER
    monkeypatch.setenv("QPN_DECODE_COOP", "2")
    cfg = PAPER
    flat = synth.make_weights(cfg, 31)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, [(61, 5, 1.0), (62, 4, 0.5)])
    m.sampling_seed = 99
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="sampling")
    o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd, mode="sampling", seed=99)
    for a, b in zip(outs, o_outs):
        np.testing.assert_array_equal(a, b)
    x, h, d, n = synth.decode_inputs(cfg, 4, 5, 1.0)
    teacher = np.random.RandomState(9).randint(0, 256, size=n).astype(np.int64)
    out, logits = m._stream_logits(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), d[None], torch.from_numpy(teacher[None]), n)
    r = oracle.decode(cfg, flat, h, d, x, n, teacher=teacher, want_logits=True)
    assert np.array_equal(logits[0].cpu().numpy().view(np.uint32), r["logits"].view(np.uint32))


def test_default_geometry_decode_vs_oracle(cuda, oracle, monkeypatch):
    """The repo-default QPNet (n_resch 512, 12 fixed + 4 adaptive layers: what runQP.py builds) decodes on the per-utterance
    cooperative kernel (QPN_DECODE_COOPB=0; the default plan's batched kernel: the tests further down): greedy streams of two
    utterances of unequal length, bit-exact vs the CPU oracle."""
    import torch
    from qpnet_amd.config import DEFAULT
    monkeypatch.setenv("QPN_DECODE_COOPB", "0")
    cfg = DEFAULT
    flat = synth.make_weights(cfg, 7)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, [(11, 2, 1.0), (12, 3, 1.5)])
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="argmax")
    assert m.last_decode_plan.startswith("coop G="), m.last_decode_plan
    o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd)
    assert [len(o) for o in outs] == sorted(ns)
    for a, b in zip(outs, o_outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("case", DECODE_CASES_D, ids=[c[0] for c in DECODE_CASES_D])
def test_default_geometry_decode_matches_reference_streams(case, cuda, golden_dir, monkeypatch):
    """the repo-default geometry against greedy streams made by the REFERENCE itself (decode_d.npz: 2 199 samples at
    B=1; B=2 of unequal lengths at F0 x 1.5), bit-exact, completion order and list consumption included -- per-utterance kernel."""
    import torch
    monkeypatch.setenv("QPN_DECODE_COOPB", "0")
    name, cfg, wseed, utts, extra = case
    g = np.load(golden_dir + "/decode_d.npz")
    m = util.build_model(cfg, synth.make_weights(cfg, wseed), cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, bd, mode="argmax")
    assert nlist == list(g[name + "_nleft"])
    assert len(outs) == len(utts)
    for i, s in enumerate(outs):
        np.testing.assert_array_equal(s, g["%s_out%d" % (name, i)].astype(np.int64), err_msg="HIP vs reference stream, row %d" % i)


# ---------------------------------------------------------------- the utterances batched into the contractions (decode_coopb.hip)
def test_batched_cooperative_decode_vs_oracle(cuda, oracle, monkeypatch):
    """Repo-default geometry, the batch as the N dimension of the fp32 MFMA (weights read once per sample step for a whole group of
    utterances): ragged lengths, three F0 factors -- greedy streams bit-exact vs the CPU oracle; then sampling mode and teacher-forced
    logits (bitwise) through the same kernel."""
    import torch
    from qpnet_amd.config import DEFAULT
    monkeypatch.delenv("QPN_DECODE_COOPB", raising=False)      # the default plan of this geometry
    cfg = DEFAULT
    flat = synth.make_weights(cfg, 19)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, [(41, 2, 1.0), (42, 1, 0.5), (43, 3, 1.5), (44, 1, 1.0), (45, 2, 0.5)])
    xb, hb = torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda)
    outs = m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
    assert m.last_decode_plan.startswith("coopb G=64 groups="), m.last_decode_plan
    o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd)
    assert [len(o) for o in outs] == sorted(ns)
    for a, b in zip(outs, o_outs):
        np.testing.assert_array_equal(a, b)
    m.sampling_seed = 77
    outs = m.batch_fast_generate(xb[:2], hb[:2], list(ns[:2]), bd[:2], mode="sampling")
    o_outs = oracle.batch_fast_generate(cfg, flat, bx[:2], bh[:2], list(ns[:2]), bd[:2], mode="sampling", seed=77)
    for a, b in zip(outs, o_outs):
        np.testing.assert_array_equal(a, b)
    x, h, d, n = synth.decode_inputs(cfg, 1, 46, 1.0)
    teacher = np.random.RandomState(3).randint(0, 256, size=n).astype(np.int64)
    out, logits = m._stream_logits(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), d[None], torch.from_numpy(teacher[None]), n)
    assert m.last_decode_plan.startswith("coopb "), m.last_decode_plan
    r = oracle.decode(cfg, flat, h, d, x, n, teacher=teacher, want_logits=True)
    assert np.array_equal(logits[0].cpu().numpy().view(np.uint32), r["logits"].view(np.uint32))


@pytest.mark.parametrize("B", [20, 37, 6, 70])
def test_batched_cooperative_decode_equals_the_per_utterance_kernel(B, cuda, monkeypatch):
    """The reference's decode batch (20, runQP.py:66) one that needs ten utterances per group, one of two launches (70 rows on a 256-CU chip): the batched kernel (the default plan)
    draws the same samples as the per-utterance cooperative kernel (QPN_DECODE_COOPB=0), which the tests above pin to the oracle
    and to the reference's own streams -- sampling mode, ragged lengths, rows finishing at different steps."""
    import torch
    from qpnet_amd.config import DEFAULT
    cfg = DEFAULT
    flat = synth.make_weights(cfg, 23)
    frames = (lambda b: 30 + 3 * b) if B == 6 else (lambda b: 4 + b % 5)      # (B = 6: up to 4 949 samples a row -- every pitch ring wraps several times)
    bx, bh, bd, ns = util.decode_batch(cfg, [(500 + b, frames(b), 0.5 + 0.25 * (b % 5)) for b in range(B)])
    xb, hb = torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda)
    res = {}
    for name, knob in (("coopb", None), ("coop", "0")):
        if knob is None:
            monkeypatch.delenv("QPN_DECODE_COOPB", raising=False)
        else:
            monkeypatch.setenv("QPN_DECODE_COOPB", knob)
        m = util.build_model(cfg, flat, cuda)
        m.sampling_seed = 11
        res[name] = (m.batch_fast_generate(xb, hb, list(ns), bd, mode="sampling"), m.last_decode_plan)
    assert res["coopb"][1].startswith("coopb ") and res["coop"][1].startswith("coop G="), (res["coopb"][1], res["coop"][1])
    for a, b in zip(res["coopb"][0], res["coop"][0]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("case", DECODE_CASES_D, ids=[c[0] for c in DECODE_CASES_D])
def test_batched_cooperative_decode_matches_reference_streams(case, cuda, golden_dir, monkeypatch):
    """decode_d.npz (greedy streams made by the REFERENCE itself on the repo-default geometry) through the batched kernel (the default plan), bit-exact."""
    import torch
    monkeypatch.delenv("QPN_DECODE_COOPB", raising=False)
    name, cfg, wseed, utts, extra = case
    g = np.load(golden_dir + "/decode_d.npz")
    m = util.build_model(cfg, synth.make_weights(cfg, wseed), cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, bd, mode="argmax")
    assert m.last_decode_plan.startswith("coopb "), m.last_decode_plan
    assert nlist == list(g[name + "_nleft"])
    for i, s_ in enumerate(outs):
        np.testing.assert_array_equal(s_, g["%s_out%d" % (name, i)].astype(np.int64), err_msg="batched HIP vs reference stream, row %d" % i)


def test_batched_cooperative_decode_other_widths_vs_oracle(cuda, oracle, monkeypatch):
    """n_resch 256 / n_skipch 512 (16-chunk current taps, 32-chunk post-net, 128-pair gathers, 16 skip and 8 logit rows per workgroup of 32),
    asked for with QPN_DECODE_COOP (one CU could hold this state): three utterances, greedy and sampling, vs the oracle; and upsampling
    factor 0 on the default widths (aux features at sample rate)."""
    import dataclasses
    import torch
    from qpnet_amd.config import QPNetConfig
    monkeypatch.setenv("QPN_DECODE_COOP", "32")
    monkeypatch.delenv("QPN_DECODE_COOPB", raising=False)
    cfg = QPNetConfig(n_resch=256, n_skipch=256, dilationF_depth=2, dilationF_repeat=1, dilationA_depth=2, dilationA_repeat=2)
    flat = synth.make_weights(cfg, 29)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, [(61, 2, 1.0), (62, 3, 0.5), (63, 1, 1.5)])
    xb, hb = torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda)
    for mode in ("argmax", "sampling"):
        m.sampling_seed = 5
        outs = m.batch_fast_generate(xb, hb, list(ns), bd, mode=mode)
        assert m.last_decode_plan.startswith("coopb G=32 "), m.last_decode_plan
        o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd, mode=mode, seed=5)
        for a, b in zip(outs, o_outs):
            np.testing.assert_array_equal(a, b)
    monkeypatch.delenv("QPN_DECODE_COOP", raising=False)
    cfg_u = QPNetConfig(dilationF_depth=2, dilationF_repeat=1, dilationA_depth=2, dilationA_repeat=1)
    cfg0 = dataclasses.replace(cfg_u, upsampling_factor=0)
    flat0 = synth.make_weights(cfg0, 31)
    m0 = util.build_model(cfg0, flat0, cuda)
    x, h, d, n = synth.decode_inputs(cfg_u, 2, 71, 1.0)
    h0 = np.ascontiguousarray(np.repeat(h, cfg_u.upsampling_factor, axis=1))          # (n_aux, T): one feature column per sample
    y = m0.batch_fast_generate(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h0[None]).to(cuda), [n], d[None], mode="argmax")[0]
    assert m0.last_decode_plan.startswith("coopb G=64 "), m0.last_decode_plan
    np.testing.assert_array_equal(y, oracle.decode(cfg0, flat0, h0, d, x, n)["samples"])


def test_batched_cooperative_decode_degenerate_lengths(cuda, oracle, monkeypatch):
    """Utterances of 0, 1 and 2 samples next to longer ones in one group of the batched kernel (rows that go inactive while their
    neighbours run on; a whole group with nothing but the warm-up to do): no hang, shortest-first retirement, the oracle's streams."""
    import torch
    from qpnet_amd.config import QPNetConfig
    monkeypatch.setenv("QPN_DECODE_COOP", "32")
    monkeypatch.delenv("QPN_DECODE_COOPB", raising=False)
    cfg = QPNetConfig(n_resch=256, n_skipch=256, dilationF_depth=2, dilationF_repeat=1, dilationA_depth=2, dilationA_repeat=1)
    flat = synth.make_weights(cfg, 37)
    m = util.build_model(cfg, flat, cuda)
    for ns in ([5, 1, 2], [1], [2], [0, 3], [130, 1, 0, 2, 7]):
        bx, bh, bd, _ = synth.decode_batch(cfg, [(100 + b, 2, 1.0) for b in range(len(ns))])
        nlist = list(ns)
        outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, bd, mode="argmax")
        assert m.last_decode_plan.startswith("coopb "), m.last_decode_plan
        refs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd)
        assert [len(o) for o in outs] == sorted(ns) == [len(r) for r in refs]
        for o, r in zip(outs, refs):
            np.testing.assert_array_equal(o, r)


def test_batched_cooperative_launch_that_gives_up_is_rerun_per_utterance(cuda):
    """tests/giveup_child.py `coopb` (the -DQPN_TESTING build's injected give-up): the batch is decoded again by decode_coop.hip."""
    util.run_giveup_child("coopb", "QPN_TEST_PIPE_GIVES_UP")


# ---------------------------------------------------------------- four pipelined workgroups per utterance, resident weights (decode_pipe.hip)
_PAPER_CASES = [c for c in DECODE_CASES if c[0].startswith("paper")]


@pytest.mark.parametrize("case", _PAPER_CASES, ids=[c[0] for c in _PAPER_CASES])
def test_pipelined_decode_matches_reference_streams(case, cuda, golden_dir, monkeypatch):
    """QPN_DECODE_PIPE=1: stack / stack / skip+post1 / post2+pick on four CUs, hand-offs by tagged granules: the same
    reference streams, bit for bit (B = 1 and B = 2 with unequal lengths, f0 x 0.5 / 1.5)."""
    import torch
    monkeypatch.setenv("QPN_DECODE_PIPE", "1")
    name, cfg, wseed, utts, extra = case
    g = np.load(golden_dir + "/decode.npz")
    m = util.build_model(cfg, synth.make_weights(cfg, wseed), cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    d_arg = torch.from_numpy(bd).float().to(cuda) if extra else bd
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, d_arg, mode="argmax", extra_memory=extra)
    for i, s in enumerate(outs):
        np.testing.assert_array_equal(s, g["%s_out%d" % (name, i)].astype(np.int64), err_msg="pipelined HIP vs reference stream, row %d" % i)


def test_pipelined_decode_worst_pitch_seeds_sampling_logits(cuda, golden_dir, oracle, monkeypatch):
    import torch
    from qpnet_amd.config import PAPER
    monkeypatch.setenv("QPN_DECODE_PIPE", "1")
    g = np.load(golden_dir + "/decode2.npz")
    for case in DECODE_CASES2:
        if case["cfg"] is not PAPER:
            continue
        cfg, name, extra = case["cfg"], case["name"], case["extra"]
        m = util.build_model(cfg, synth.make_weights(cfg, case["wseed"]), cuda)
        bx, bh, bd, ns = decode2_inputs(case)
        d_arg = torch.from_numpy(bd).float().to(cuda) if extra else bd
        outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), d_arg, mode="argmax", extra_memory=extra)
        for i, s in enumerate(outs):
            np.testing.assert_array_equal(s, g["%s_out%d" % (name, i)].astype(np.int64), err_msg=name)
    cfg = PAPER
    flat = synth.make_weights(cfg, 31)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, [(61 + b, 4 + (b % 3), 1.0) for b in range(11)])        # more than one 8-utterance chunk
    m.sampling_seed = 4242
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="sampling")
    o_outs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd, mode="sampling", seed=4242)
    for a, b in zip(outs, o_outs):
        np.testing.assert_array_equal(a, b)
    x, h, d, n = synth.decode_inputs(cfg, 4, 5, 1.0)
    teacher = np.random.RandomState(9).randint(0, 256, size=n).astype(np.int64)
    out, logits = m._stream_logits(torch.from_numpy(x[None]).to(cuda), torch.from_numpy(h[None]).to(cuda), d[None], torch.from_numpy(teacher[None]), n)
    r = oracle.decode(cfg, flat, h, d, x, n, teacher=teacher, want_logits=True)
    assert np.array_equal(logits[0].cpu().numpy().view(np.uint32), r["logits"].view(np.uint32))


@pytest.mark.parametrize("kernel", ["pipelined", "one_cu", "cooperative"])
def test_degenerate_lengths_on_every_kernel(kernel, cuda, oracle, monkeypatch):
    """Utterances of 0, 1 and 2 samples next to longer ones (the persistent loops have nothing, or only the warm-up step,
    to run for them): no hang, shortest-first retirement order, every stream equal to the oracle's
    (reference qpnet.py:417-419, 521-559)."""
    import torch
    from qpnet_amd.config import PAPER
    if kernel == "one_cu":
        monkeypatch.setenv("QPN_DECODE_PIPE", "0")
    elif kernel == "cooperative":
        monkeypatch.setenv("QPN_DECODE_COOP", "4")
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    m = util.build_model(cfg, flat, cuda)
    for ns in ([5, 1, 2], [1], [2], [0, 3], [330, 1]):
        bx, bh, bd, _ = synth.decode_batch(cfg, [(100 + b, 3, 1.0) for b in range(len(ns))])
        nlist = list(ns)
        outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, bd, mode="argmax")
        refs = oracle.batch_fast_generate(cfg, flat, bx, bh, list(ns), bd)
        assert [len(o) for o in outs] == sorted(ns) == [len(r) for r in refs]
        for o, r in zip(outs, refs):
            np.testing.assert_array_equal(o, r)


def test_pipelined_decode_at_full_occupancy(cuda, oracle):
    """48 ragged utterances in one call: 240 of the 256 CUs hold the five resident roles of an utterance each (the largest batch the
    pipelined kernel takes).  Mixed F0 scalings, lengths 40..99 frames; six rows against their single-row oracle streams, the
    whole batch repeatable, shortest-first order kept."""
    import torch
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    m = util.build_model(cfg, flat, cuda)
    specs = [(100 + b, 40 + (b * 7) % 60, [0.5, 1.0, 1.5][b % 3]) for b in range(48)]
    bx, bh, bd, ns = synth.decode_batch(cfg, specs)
    xb, hb = torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda)
    outs = m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
    assert m.last_decode_kernel_ms > 0
    order = np.argsort(ns, kind="stable")
    assert [len(o) for o in outs] == [ns[b] for b in order]
    for k in (0, 7, 19, 30, 41, 47):
        b = int(order[k])
        x, h, d, n = synth.decode_inputs(cfg, specs[b][1], specs[b][0], specs[b][2])
        maxd = int(np.ceil(np.nanmax(bd)))
        np.testing.assert_array_equal(outs[k], oracle.decode(cfg, flat, h, d, x, n, maxd=maxd)["samples"], err_msg="row %d" % b)
    outs2 = m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
    for a, b2 in zip(outs, outs2):
        np.testing.assert_array_equal(a, b2)


# ---------------------------------------------------------------- launch plans beyond one pipelined launch (round 3)
def _paper_batch(B, lo=6, span=9, seed0=300):
    from qpnet_amd.config import PAPER
    specs = [(seed0 + b, lo + (b * 5) % span, [1.0, 0.5, 1.5][b % 3]) for b in range(B)]
    return PAPER, specs


@pytest.mark.parametrize("B,mode", [(49, "argmax"), (64, "sampling"), (100, "argmax")])
def test_batches_beyond_one_pipelined_launch(B, mode, cuda, oracle):
    """decode_batch_size is a free parameter of the reference (--batch_size, src/bin/qpnet_decode.py:52).  More rows than the 48
    five-role groups a 256-CU device holds resident: groups take a second utterance, stepped alternately, or a third (97-144 rows:
    one launch of three per group is cheaper than two launches); whatever the plan, EVERY row equals its single-row oracle stream
    (sampling: the Philox key is the caller's row number).  Plans of SEVERAL launches: test_two_launch_plan below."""
    import torch
    cfg, specs = _paper_batch(B)
    flat = synth.make_weights(cfg, 13)
    m = util.build_model(cfg, flat, cuda)
    m.sampling_seed = 4242
    bx, bh, bd, ns = synth.decode_batch(cfg, specs)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode=mode)
    import re
    mt = re.match(r"pipe rows=(\d+) waves=(\d+) x (\d+) \((\d) per group\); one-cu rows=(\d+)", m.last_decode_plan)
    assert mt, m.last_decode_plan
    n_pipe, n_waves, per, per_group, n_one = map(int, mt.groups())
    assert n_pipe == B and n_one == 0 and n_waves * per >= n_pipe, m.last_decode_plan
    if B in (49, 64):
        assert n_waves == 1 and per_group == 2                      # one launch: 48 groups, B - 48 of them with two utterances
    if B == 100:
        assert n_waves == 1 and per_group == 3                      # three per group saves the second launch (15.3 us a step against 2 x 10)
    order = np.argsort(ns, kind="stable")
    assert [len(o) for o in outs] == [ns[b] for b in order]
    maxd = int(np.ceil(np.nanmax(bd)))
    for k in range(B):
        b = int(order[k])
        x, h, d, n = synth.decode_inputs(cfg, specs[b][1], specs[b][0], specs[b][2])
        r = oracle.decode(cfg, flat, h, d, x, n, maxd=maxd, mode=mode, seed=4242, row=b)
        np.testing.assert_array_equal(outs[k], r["samples"], err_msg="row %d (%s)" % (b, m.last_decode_plan))


@pytest.mark.parametrize("mode", ["argmax", "sampling"])
def test_two_launch_plan(mode, cuda, oracle, monkeypatch):
    """A plan of TWO pipelined launches (descriptor slices d_utts + first, the exchange block cleared per launch, ring offsets keyed by the
    sorted position): with at most two utterances per group (QPN_PIPE_NU=2, read at qpn_create) 100 rows exceed the 96 one launch
    holds on a 256-CU device.  Every row equals its single-row oracle stream."""
    import re
    import torch
    monkeypatch.setenv("QPN_PIPE_NU", "2")
    B = 100
    cfg, specs = _paper_batch(B, lo=5, span=5, seed0=900)
    flat = synth.make_weights(cfg, 13)
    m = util.build_model(cfg, flat, cuda)               # (a new module: its handle is created under the knob)
    m.sampling_seed = 777
    bx, bh, bd, ns = synth.decode_batch(cfg, specs)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode=mode)
    mt = re.match(r"pipe rows=(\d+) waves=(\d+) x (\d+) \((\d) per group\); one-cu rows=(\d+)", m.last_decode_plan)
    assert mt, m.last_decode_plan
    n_pipe, n_waves, per, per_group, n_one = map(int, mt.groups())
    if torch.cuda.get_device_properties(cuda).multi_processor_count == 256:
        assert n_waves == 2 and per_group == 2 and n_pipe == B and n_one == 0, m.last_decode_plan
    order = np.argsort(ns, kind="stable")
    assert [len(o) for o in outs] == [ns[b] for b in order]
    maxd = int(np.ceil(np.nanmax(bd)))
    for k in range(B):
        b = int(order[k])
        x, h, d, n = synth.decode_inputs(cfg, specs[b][1], specs[b][0], specs[b][2])
        r = oracle.decode(cfg, flat, h, d, x, n, maxd=maxd, mode=mode, seed=777, row=b)
        np.testing.assert_array_equal(outs[k], r["samples"], err_msg="row %d (%s)" % (b, m.last_decode_plan))


def test_equal_length_rows_beyond_capacity_share_groups(cuda, oracle):
    """80 rows of one length: 32 of the 48 resident groups take a second utterance (one launch)."""
    import torch
    from qpnet_amd.config import PAPER
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    m = util.build_model(cfg, flat, cuda)
    specs = [(500 + b % 4, 8, 1.0) for b in range(80)]
    bx, bh, bd, ns = synth.decode_batch(cfg, specs)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), list(ns), bd, mode="argmax")
    assert "pipe rows=80 waves=1 x 80 (2 per group); one-cu rows=0" in m.last_decode_plan, m.last_decode_plan
    ref = {}
    for k in range(80):
        key = specs[k][0]
        if key not in ref:
            x, h, d, n = synth.decode_inputs(cfg, 8, key, 1.0)
            ref[key] = oracle.decode(cfg, flat, h, d, x, n, maxd=int(np.ceil(np.nanmax(bd))))["samples"]
        np.testing.assert_array_equal(outs[k], ref[key])


def test_pipelined_launch_that_gives_up_is_rerun_on_one_cu_kernels(cuda):
    """a multi-workgroup launch whose workgroups are not co-resident (CU-masked / shared GPU) times out and drains; the call
    is then re-run on the one-CU kernel instead of failing with QPN_ENODEV.  The give-up is injected by a hook that only the
    -DQPN_TESTING build of the library contains: tests/giveup_child.py `pipe`, in a child process bound to that build."""
    util.run_giveup_child("pipe", "QPN_TEST_PIPE_GIVES_UP")


def test_enqueue_is_asynchronous_and_single_flight(cuda):
    """qpn_decode_enqueue returns without synchronising (descriptors staged in pinned memory); a second enqueue before
    qpn_decode_finish is refused."""
    import ctypes as C
    import torch
    from qpnet_amd import _lib
    from qpnet_amd.config import PAPER
    cfg = PAPER
    m = util.build_model(cfg, synth.make_weights(cfg, 13), cuda)
    L, hd = m._native(cuda)
    x, h, d, n = synth.decode_inputs(cfg, 60, 5, 1.0)
    xt = torch.from_numpy(x[None]).to(cuda); ht = torch.from_numpy(h[None]).to(cuda); dt = torch.from_numpy(d[None]).to(cuda)
    out = torch.empty((1, n), dtype=torch.int64, device=cuda)
    stream = torch.cuda.current_stream(cuda).cuda_stream
    m._bind_decode_weights(L, hd, cuda, stream)
    arr = (C.c_int64 * 1)(n)
    maxd = int(np.ceil(d.max()))
    torch.cuda.synchronize()
    args = (hd, 1, 1, h.shape[1], d.shape[0], xt.data_ptr(), ht.data_ptr(), dt.data_ptr(), 0, arr, maxd, 0, 0, None, out.data_ptr(), None, stream)
    _lib.check(L.qpn_decode_enqueue(*args))
    busy = not torch.cuda.current_stream(cuda).query()            # 6.6 k samples at >= 7 us each: still running when enqueue returns
    rc = L.qpn_decode_enqueue(*args)
    assert rc == -5 and b"in flight" in L.qpn_last_error()
    _lib.check(L.qpn_decode_finish(hd, stream))
    assert busy, "qpn_decode_enqueue blocked until the decode had finished"
    assert int(out[0, -1]) >= 0


def test_deep_network_decode_matches_reference_stream(cuda, golden_dir, oracle):
    """The reference's second shipped network, 'Rd10Rr3Ed4Er1' (src/utils/param_model.py:66-72: dilationF_depth 10 x repeat 3 + 4 adaptive = 34 layers,
    fixed dilations up to 512), at the paper-size widths: the HIP decode against the greedy stream made by the REFERENCE itself (decode_deep.npz,
    2 199 samples), bit-exact -- and against the oracle with teacher-forced logits, bit for bit."""
    import torch
    from cases import DECODE_CASES_DEEP
    name, cfg, wseed, utts, extra = DECODE_CASES_DEEP[0]
    g = np.load(golden_dir + "/decode_deep.npz")
    flat = synth.make_weights(cfg, wseed)
    m = util.build_model(cfg, flat, cuda)
    bx, bh, bd, ns = util.decode_batch(cfg, utts)
    nlist = list(ns)
    outs = m.batch_fast_generate(torch.from_numpy(bx).to(cuda), torch.from_numpy(bh).to(cuda), nlist, bd, mode="argmax")
    assert nlist == list(g[name + "_nleft"]) and len(outs) == 1
    np.testing.assert_array_equal(outs[0], g[name + "_out0"].astype(np.int64), err_msg="HIP vs reference stream")
