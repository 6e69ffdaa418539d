"""CPU: host-side counterparts of the reference batch generators."""
import numpy as np
import torch

from qpnet_amd import loaders, synth, harness
from qpnet_amd.config import PAPER, TINY


def test_decode_generator_semantics():
    cfg = PAPER
    lens = [12, 5, 9, 5, 20]
    feats = [synth.make_features(n, 100 + i) for i, n in enumerate(lens)]
    mean, scale = synth.scaler_stats()
    batches = list(loaders.decode_generator(feats, 22050, wav_transform=loaders.mu_law_transform(256),
                                            feat_transform=lambda h: (h - mean) / scale, batch_size=2,
                                            upsampling_factor=cfg.upsampling_factor, f0_factor=1.5))
    assert len(batches) == 3                                   # ceil(5/2) batches via array_split -> sizes 2,2,1
    ids, bx, bh, ns, bd = batches[0]
    assert ids == ["utt0001", "utt0003"] and ns == [5 * 110 - 1, 5 * 110 - 1]     # stable sort by length
    assert bx.dtype == torch.int64 and bx.shape == (2, 1) and int(bx[0, 0]) == 128  # mu-law of 0
    assert bh.shape == (2, 39, 5) and bd.shape == (2, 550) and bd.dtype == np.float64
    # F0 scaled BEFORE d and before normalisation: d = fs / (1.5 f0 * 8)
    np.testing.assert_allclose(bd[0, ::110], 22050.0 / (feats[1][:, 1].astype(np.float64) * 1.5) / 8.0, rtol=1e-6)
    ids2, _, bh2, ns2, bd2 = batches[1]
    assert ns2 == [9 * 110 - 1, 12 * 110 - 1] and bh2.shape[2] == 12
    assert float(bh2[0, :, 9:].abs().max()) == 0.0 and float(np.abs(bd2[0, 990:]).max()) == 0.0   # zero padding


def test_train_generator_chunks():
    cfg = TINY
    U = cfg.upsampling_factor
    rs = np.random.RandomState(0)
    utts = []
    for i in range(3):
        h = synth.make_features(60, 200 + i)
        utts.append((rs.uniform(-1, 1, 60 * U + 7).astype(np.float32), h))
    gen = loaders.train_generator(utts, cfg.receptiveCausal_field, cfg.receptiveF_field, cfg.receptiveA_field, 22050,
                                  wav_transform=loaders.mu_law_transform(256), batch_length=1500, max_length=4000,
                                  upsampling_factor=U, shuffle=False, epochs=1)
    chunks = list(gen)
    assert len(chunks) >= 5
    x0, h0, t0, d0, b0 = chunks[0]
    bl = int(b0[0])
    T = x0.shape[1]
    assert T % U == 0 and h0.shape == (1, 39, T // U) and d0.shape == (1, T)
    assert torch.equal(x0[0, 1:], t0[0, :-1])                       # targets = inputs shifted by one
    # first chunk = first samples of the first utterance, mu-law encoded
    from qpnet_amd.qpnet import encode_mu_law
    np.testing.assert_array_equal(x0[0].numpy(), encode_mu_law(utts[0][0][:T], 256))
    # consecutive chunks advance by batch_length rounded down to whole frames (they overlap by >= the receptive field)
    x1 = chunks[1][0]
    shift = (bl // U) * U
    ov = T - shift
    assert ov >= T - bl > 0 and torch.equal(x0[0, shift:], x1[0, :ov])
    # geometry helper agrees
    d_all = harness.extend_time(harness.dilated_factor(harness.batch_f0(utts[0][1], 0), 22050, 8)[:, None], U)[:, 0]
    rf, bl2, h_bs, x_bs = harness.train_chunk_geometry(cfg, d_all, 1500, 4000)
    assert bl2 == bl and x_bs - 1 == T


def test_checkpoint_roundtrip(tmp_path):
    from qpnet_amd.qpnet import QPNet, initialize
    m = QPNet(**TINY.kwargs()); m.apply(initialize)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    p = loaders.save_checkpoint(str(tmp_path), m, opt, 1234)
    ck = torch.load(p, weights_only=False)
    assert set(ck.keys()) == {"model", "optimizer", "iterations"} and ck["iterations"] == 1234
    assert list(ck["model"].keys()) == [k for k, _ in TINY.param_layout()]
    m2 = QPNet(**TINY.kwargs())
    assert loaders.load_checkpoint(p, m2, torch.optim.Adam(m2.parameters(), lr=1e-4)) == 1234
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    pf = loaders.save_final(str(tmp_path), m)
    assert set(torch.load(pf, weights_only=False).keys()) == {"model"}


def test_wav_writer_matches_reference_scaling(tmp_path):
    from scipy.io import wavfile
    ids = np.array([0, 16, 128, 239, 255], dtype=np.int64)
    pcm = loaders.samples_to_int16(ids)
    # decode_mu_law KATs: 0 -> -1.02207017 (clipped), 128 -> 0, 255 -> 0.97840458
    assert pcm.dtype == np.int16 and pcm[0] == -32768 and pcm[2] == 0 and pcm[4] == int(0.97840458 * 32768)
    p = loaders.write_wav(str(tmp_path / "out" / "a.wav"), 22050, ids)
    fs, back = wavfile.read(p)
    assert fs == 22050 and np.array_equal(back, pcm)


def test_hdf5_helpers_fail_loudly_without_h5py():
    import importlib.util, pytest
    if importlib.util.find_spec("h5py") is not None:
        pytest.skip("h5py present")
    with pytest.raises(ImportError):
        loaders.read_hdf5("nope.h5", "/world")
