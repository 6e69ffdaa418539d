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


def test_hdf5_helpers_fail_loudly_without_any_hdf5(monkeypatch):
    """no h5py and no HDF5 C library: the .h5 helpers raise an ImportError that names the way out (no silent stand-in); a missing file is its own error"""
    import builtins, pytest
    from qpnet_amd import _hdf5
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == "h5py":
            raise ImportError("no h5py (test)")
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", no_h5py)

    def no_lib():
        raise ImportError("no HDF5 library found (test)")
    monkeypatch.setattr(_hdf5, "_lib", no_lib)
    with pytest.raises(ImportError, match="QPN_LIBHDF5"):
        loaders.read_hdf5("nope.h5", "/world")
    monkeypatch.undo()
    with pytest.raises((FileNotFoundError, ImportError)):
        loaders.read_hdf5("nope.h5", "/world")


def test_train_generator_shards_are_the_round_robin_of_the_full_stream():
    """data-parallel ranks walk the same stream; rank r materialises batches r, r+N, ... only (runners._batches)."""
    cfg = TINY
    U = cfg.upsampling_factor
    rs = np.random.RandomState(3)
    utts = [(rs.uniform(-1, 1, 70 * U + 3).astype(np.float32), synth.make_features(70, 300 + i)) for i in range(4)]
    calls = {"n": 0}

    def counting_mu_law(x):
        calls["n"] += 1
        return loaders.mu_law_transform(256)(x)

    def run(shard):
        np.random.seed(7)                       # same seed on every rank: identical shuffles
        calls["n"] = 0
        gen = loaders.train_generator(utts, cfg.receptiveCausal_field, cfg.receptiveF_field, cfg.receptiveA_field, 22050,
                                      wav_transform=counting_mu_law, batch_length=1500, max_length=4000,
                                      upsampling_factor=U, shuffle=True, epochs=2, shard=shard)
        return list(gen), calls["n"]
    full, n_full = run(None)
    assert len(full) >= 8 and n_full == len(full)
    for world in (2, 3):
        total = 0
        for rank in range(world):
            part, n_part = run((rank, world))
            assert n_part == len(part)          # nothing is encoded for a chunk another rank consumes
            want = full[rank::world]
            assert len(part) == len(want)
            for a, b in zip(part, want):
                assert all(torch.equal(u, v) for u, v in zip(a, b))
            total += len(part)
        assert total == len(full)


def test_sharded_generator_loads_only_the_utterances_its_chunks_reach(golden_dir):
    """with world 8 a rank used to load (and validate, and take the dilated factors of) EVERY utterance of the stream.  Now the chunk
    plan comes from per-utterance metadata -- the `plan()` hook of runners' file-backed loaders (wav header + features, no waveform), cached over the epochs --
    and an utterance is loaded when one of the rank's own chunks reaches into it: paper-size receptive fields, 48 short utterances, 3 epochs (144 loads at
    world 1): each of the 8 ranks loads no more than a world-th of that plus what its chunks' overlap brings in, the shards are still the round robin of
    the full stream, and without the hook only the FIRST epoch loads everything (the metadata is cached)."""
    from qpnet_amd.config import PAPER
    cfg = PAPER
    U = cfg.upsampling_factor
    rs = np.random.RandomState(3)
    N, EPOCHS = 48, 3
    data = [(rs.uniform(-1, 1, nf * U + 3).astype(np.float32), synth.make_features(nf, 300 + i, 45.0, 300.0)) for i, nf in enumerate(rs.randint(40, 90, size=N))]

    class Utt:
        def __init__(self, i):
            self.i, self.calls, self.plans = i, 0, 0

        def __call__(self):
            self.calls += 1
            return data[self.i]

    class UttWithPlan(Utt):
        def plan(self):
            self.plans += 1
            return len(data[self.i][0]), data[self.i][1]

    class UttWithPart(UttWithPlan):                              # ... and a source that hands out slices (runners._FileUtterance: memory maps)
        def part(self, s0, s1, f0, f1):
            self.parts = getattr(self, "parts", 0) + 1
            return data[self.i][0][s0:s1], data[self.i][1][f0:f1]

    def run(cls, shard):
        utts = [cls(i) for i in range(N)]
        np.random.seed(7)
        gen = loaders.train_generator(utts, cfg.receptiveCausal_field, cfg.receptiveF_field, cfg.receptiveA_field, 22050,
                                      wav_transform=loaders.mu_law_transform(256), batch_length=20000, max_length=30000,
                                      upsampling_factor=U, shuffle=True, epochs=EPOCHS, shard=shard)
        out = list(gen)
        assert loaders.train_generator.last_stats["loads"] == sum(u.calls for u in utts)
        return out, sum(u.calls for u in utts), sum(u.plans for u in utts)
    full, loads1, _ = run(Utt, None)
    assert loads1 == N * EPOCHS and len(full) >= 40
    frames_per_chunk = full[0][1].shape[2]
    reach = int(np.ceil(frames_per_chunk / 40.0)) + 1            # utterances a chunk can reach into (shortest utterance: 40 frames)
    for cls in (UttWithPart, UttWithPlan, Utt):
        total = 0
        for rank in range(8):
            part, loads, plans = run(cls, (rank, 8))
            want = full[rank::8]
            assert len(part) == len(want) and all(all(torch.equal(u, v) for u, v in zip(a, b)) for a, b in zip(part, want))
            total += len(part)
            if cls is UttWithPart:
                st = loaders.train_generator.last_stats
                assert plans == N and loads == 0 and 0 < st["parts"] <= len(part) * reach      # slices only: a world-th of the bytes
            elif cls is UttWithPlan:
                assert plans == N                                # the plan of every utterance, once (not once per epoch)
                assert loads <= len(part) * reach and loads <= loads1 // 8 + len(part) * 2, (rank, loads, len(part))
            else:
                assert loads <= N + len(part) * reach            # first epoch: everything once (that IS the plan); later epochs: its own only
        assert total == len(full)


# ---------------------------------------------------------------- pinned to the reference's own generators (generators.npz)
def _sk_scaler(mean, scale):
    from sklearn.preprocessing import StandardScaler
    sc = StandardScaler()
    sc.mean_, sc.scale_ = mean, scale
    return sc.transform


import pytest                                                                      # noqa: E402
from cases import GEN_TRAIN_CASES, GEN_DECODE_CASES, generator_corpus, crc        # noqa: E402


@pytest.mark.parametrize("case", GEN_TRAIN_CASES, ids=[c["name"] for c in GEN_TRAIN_CASES])
def test_train_generator_equals_reference_generator(case, golden_dir):
    """chunk for chunk: shapes, batch_length_current and the bytes of x / h / t / d equal what the reference's
    train_generator (src/bin/qpnet_train.py:200-335) yielded on the same utterances, incl. a wav shorter than its
    frames, batch_length shrunk by max_length, per-chunk receptive fields and the epoch wrap with re-shuffles."""
    gold = np.load(golden_dir + "/generators.npz")["train_" + case["name"]]
    pcm, feats, mean, scale = generator_corpus(case["corpus_seed"], case["frames"], case["sample_slack"], case["f0"], case["U"])
    utts = [(p.astype(np.float32) / 32768, f) for p, f in zip(pcm, feats)]
    np.random.seed(case["np_seed"])
    gen = loaders.train_generator(utts, case["rc"], case["rf"], case["ra"], 22050, wav_transform=loaders.mu_law_transform(256),
                                  feat_transform=_sk_scaler(mean, scale), dense_factor=8, batch_length=case["batch_length"],
                                  batch_size=case["batch_size"], max_length=case["max_length"], f0_threshold=case["f0_threshold"],
                                  upsampling_factor=case["U"], shuffle=case["shuffle"])
    for k in range(case["n_batches"]):
        bx, bh, bt, bd, bb = next(gen)
        assert bx.dtype == torch.int64 and bh.dtype == torch.float32 and bd.dtype == torch.float32
        row = [bx.shape[0], bx.shape[1], bh.shape[2], int(bb[0]), crc(bx.numpy()), crc(bh.numpy()), crc(bt.numpy()),
               crc(bd.numpy()), int(bb.sum())]
        assert row == gold[k].tolist(), "batch %d differs from the reference generator's" % k


@pytest.mark.parametrize("case", GEN_DECODE_CASES, ids=[c["name"] for c in GEN_DECODE_CASES])
def test_decode_generator_equals_reference_generator(case, golden_dir):
    """batch composition, order, n_samples_list and the bytes of x / h / d equal the reference's decode_generator
    (src/bin/qpnet_decode.py:122-209)."""
    g = np.load(golden_dir + "/generators.npz")
    gold = g["decode_" + case["name"]]
    _, feats, mean, scale = generator_corpus(case["corpus_seed"], case["frames"], None, case["f0"], case["U"])
    ids = ["%d" % i for i in range(len(feats))]
    gen = loaders.decode_generator(feats, 22050, ids, wav_transform=loaders.mu_law_transform(256),
                                   feat_transform=_sk_scaler(mean, scale), dense_factor=8, batch_size=case["batch_size"],
                                   upsampling_factor=case["U"], f0_factor=float(str(case["f0_factor"])), f0_dim_index=1,
                                   extra_memory=case["extra_memory"])
    order, ns_all = [], []
    for k, (feat_ids, bx, bh, ns, bd) in enumerate(gen):
        bdn = bd.numpy() if case["extra_memory"] else bd
        assert bdn.dtype == (np.float32 if case["extra_memory"] else np.float64)
        row = [len(feat_ids), bx.shape[1], bh.shape[2], bdn.shape[1], crc(bx.numpy()), crc(bh.numpy()), crc(bdn)]
        assert row == gold[k].tolist(), "decode batch %d differs from the reference generator's" % k
        order += [int(s) for s in feat_ids]; ns_all += list(ns)
    assert k + 1 == len(gold)
    assert order == g["decode_" + case["name"] + "_order"].tolist()
    assert ns_all == g["decode_" + case["name"] + "_ns"].tolist()
    # the caller's feature arrays are not modified by the F0 scaling
    _, feats2, _, _ = generator_corpus(case["corpus_seed"], case["frames"], None, case["f0"], case["U"])
    assert all(np.array_equal(a, b) for a, b in zip(feats, feats2))


def test_sample_window_max_factor_ignores_nan_like_nanmax():
    """The reference takes np.nanmax over the live dilated factors (src/bin/qpnet_train.py _receptive_field): a NaN factor is skipped,
    it does not hide the segment it sits in."""
    from qpnet_amd.loaders import _SampleWindow
    U = 4
    w = _SampleWindow(3, np.float32, U)
    d1 = np.array([5.0, np.nan, 9.0, 2.0]); d2 = np.array([np.nan, 7.0, 3.0])
    w.append(np.zeros(len(d1) * U), np.zeros((len(d1), 3), np.float32), d1)
    w.append(np.zeros(len(d2) * U), np.zeros((len(d2), 3), np.float32), d2)
    assert w.max_factor() == np.nanmax(np.concatenate([d1, d2])) == 9.0
    w.f0 = 3                                             # the first three frames of the first segment are consumed
    assert w.max_factor() == np.nanmax(np.concatenate([d1[3:], d2])) == 7.0
