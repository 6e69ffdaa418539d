"""GPU: the task entry points (python -m qpnet_amd.run_train|run_update|run_validate|run_decode) on a small synthetic
corpus of .wav + feature files -- .npy, and the reference's own format: HDF5 files with a /world dataset and a stats file with /world/mean and
/world/scale (src/utils/utils.py:43-128, calc_stats.py), read through h5py or libhdf5 --: stages 1-3 of run_QP.sh on the native path, incl. checkpoint resume."""
import argparse
import os

import numpy as np
import pytest
import yaml

from qpnet_amd import loaders, synth
from qpnet_amd.config import TINY

pytestmark = pytest.mark.gpu


def _corpus(root, n=3, frames=45, fmt="npy"):
    from scipy.io import wavfile
    U = TINY.upsampling_factor
    os.makedirs(root + "/wav"); os.makedirs(root + "/feat")
    rs = np.random.RandomState(5)
    feats = []
    for i in range(n):
        h = synth.make_features(frames + 3 * i, 700 + i)
        x = (rs.uniform(-0.8, 0.8, (frames + 3 * i) * U + 11) * 32767).astype(np.int16)
        wavfile.write("%s/wav/u%02d.wav" % (root, i), 22050, x)
        if fmt == "h5":
            loaders.write_hdf5("%s/feat/u%02d.h5" % (root, i), "/world", h)       # what feature_extract.py writes (:337-343)
        else:
            np.save("%s/feat/u%02d.npy" % (root, i), h)
        feats.append(h)
    st = loaders.calc_stats(feats)
    if fmt == "h5":
        loaders.write_hdf5(root + "/stats.h5", "/world/mean", st.mean_)          # calc_stats.py:33-36
        loaders.write_hdf5(root + "/stats.h5", "/world/scale", st.scale_)
    else:
        np.savez(root + "/stats.npz", mean=st.mean_, scale=st.scale_)
    return root


@pytest.mark.parametrize("fmt", ["npy", "h5"])
def test_train_update_validate_decode_entry_points(fmt, cuda, tmp_path, oracle):
    import torch
    from qpnet_amd import runners
    if fmt == "h5":
        try:
            loaders._h5py()
        except ImportError:
            pytest.skip("neither h5py nor libhdf5 here")
    root = _corpus(str(tmp_path / "corpus"), fmt=fmt)
    exp = str(tmp_path / "exp")
    stats = root + ("/stats.h5" if fmt == "h5" else "/stats.npz")
    common = ["--waveforms", root + "/wav", "--feats", root + "/feat", "--stats", stats]
    geo = ["--n_resch", "32", "--n_skipch", "32", "--dilationF_depth", "2", "--dilationF_repeat", "1", "--dilationA_depth", "1",
           "--dilationA_repeat", "1", "--feature_format", fmt, "--batch_length", "1500", "--max_length", "4000", "--verbose", "0"]
    conf = exp + "/model.conf"
    assert runners.run_train(common + geo + ["--expdir", exp, "--config", conf, "--iters", "6", "--checkpoint_interval", "3",
                                             "--intervals", "2", "--resume", exp + "/none.pkl"]) == 0
    assert os.path.exists(exp + "/checkpoint-3.pkl") and os.path.exists(exp + "/checkpoint-6.pkl") and os.path.exists(exp + "/checkpoint-final.pkl")
    rec = yaml.safe_load(open(exp + "/loss-final.yml"))
    assert len(rec) == 3 and all(np.isfinite(rec))
    assert isinstance(loaders.load_model_conf(conf), argparse.Namespace)
    # resume from iteration 3: runs 3 more steps and ends with the same weights as the uninterrupted run (same seed -> same stream)
    exp2 = str(tmp_path / "exp2")
    os.makedirs(exp2)
    assert runners.run_train(common + geo + ["--expdir", exp2, "--config", exp2 + "/model.conf", "--iters", "6", "--checkpoint_interval", "3",
                                             "--intervals", "2", "--resume", exp + "/checkpoint-3.pkl"]) == 0
    a = torch.load(exp + "/checkpoint-final.pkl", map_location="cpu")["model"]; b = torch.load(exp2 + "/checkpoint-final.pkl", map_location="cpu")["model"]
    # the resumed run restarts the shuffled stream from its beginning (as the reference does), so only shapes/finite-ness are comparable
    assert a.keys() == b.keys() and all(torch.isfinite(v).all() for v in b.values())
    # SD update from the SI model
    exp3 = str(tmp_path / "sd")
    assert runners.run_update(common + ["--expdir", exp3, "--config", conf, "--pretrain", exp + "/checkpoint-final.pkl", "--iters", "2",
                                        "--batch_length", "1500", "--max_length", "4000", "--intervals", "1", "--verbose", "0",
                                        "--resume", exp3 + "/none.pkl"]) == 0
    assert os.path.exists(exp3 + "/checkpoint-final.pkl")
    # validation result file
    res = str(tmp_path / "val")
    assert runners.run_validate(common + ["--resultdir", res, "--config", conf, "--checkpoint", exp + "/checkpoint-final.pkl",
                                          "--batch_length", "1500", "--max_length", "4000", "--verbose", "0"]) == 0
    val = yaml.safe_load(open(res + "/validation_result.yml"))
    assert list(val) == ["checkpoint-final.pkl"] and 0 < val["checkpoint-final.pkl"] < 10
    # ... and the number itself: the same batches through the numpy oracle (the staged batches reach forward_loss on a copy stream; an
    # un-joined copy would show as a loss computed on stale inputs)
    assert abs(val["checkpoint-final.pkl"] - _oracle_validation_loss(root, fmt, stats, conf, exp + "/checkpoint-final.pkl", 1500, 4000)) < 1e-4
    # decode (greedy so the oracle can check it), wav files named by feature id
    out = str(tmp_path / "wav_out")
    assert runners.run_decode(["--feats", root + "/feat", "--stats", stats, "--config", conf, "--checkpoint",
                               exp + "/checkpoint-final.pkl", "--outdir", out + "/feat_id.wav", "--batch_size", "2", "--mode", "argmax",
                               "--intervals", "2000", "--verbose", "0"]) == 0
    from scipy.io import wavfile
    sd = torch.load(exp + "/checkpoint-final.pkl", map_location="cpu")["model"]
    flat = np.concatenate([v.numpy().ravel() for v in sd.values()]).astype(np.float32)
    sc = loaders.read_scaler_stats(stats)
    feat = lambda i: loaders.read_features("%s/feat/u%02d.%s" % (root, i, fmt))
    for i in range(3):
        fs, w = wavfile.read("%s/u%02d.wav" % (out, i))
        h = feat(i)
        assert fs == 22050 and w.dtype == np.int16 and len(w) == h.shape[0] * TINY.upsampling_factor - 1
    # the shortest utterance re-decoded by the oracle from the checkpoint == the wav that was written
    h = feat(0)
    from qpnet_amd import harness
    d = harness.extend_time(harness.dilated_factor(harness.batch_f0(h), 22050, 8)[:, None], TINY.upsampling_factor)[:, 0]
    # batch-level maxd: u00 was decoded together with u01 (batch_size 2, sorted by length)
    h1 = feat(1)
    d1 = harness.dilated_factor(harness.batch_f0(h1), 22050, 8)
    maxd = int(np.ceil(max(d.max(), d1.max())))
    hn = sc.transform(h).astype(np.float32)
    ref = oracle.decode(TINY, flat, np.ascontiguousarray(hn.T), d, np.array([128], dtype=np.int64), h.shape[0] * 110 - 1, maxd=maxd)["samples"]
    np.testing.assert_array_equal(wavfile.read(out + "/u00.wav")[1], loaders.samples_to_int16(ref))


def _oracle_validation_loss(root, fmt, stats, conf_path, ckpt, batch_length, max_length):
    """run_validate's number restated on the host: the reference's validation pass (qpnet_validate.py:409-437) = mean over the un-shuffled generator's
    batches of the mean CE of each, here through oracle/train_oracle.py on the checkpoint's weights."""
    import torch
    from oracle import train_oracle as TO
    conf = loaders.load_model_conf(conf_path)
    wavs, feats = loaders.file_lists(root + "/wav", root + "/feat", fmt)
    scaler = loaders.read_scaler_stats(stats, conf.feature_type)
    fs = loaders.read_wav(wavs[0])[0]
    items = [(lambda w=w, f=f: (loaders.read_wav(w)[1], loaders.read_features(f, conf.feature_type))) for w, f in zip(wavs, feats)]
    gen = loaders.train_generator(items, TINY.receptiveCausal_field, TINY.receptiveF_field, TINY.receptiveA_field, fs,
                                  wav_transform=loaders.mu_law_transform(conf.n_quantize), feat_transform=scaler, dense_factor=conf.dense_factor,
                                  batch_length=batch_length, batch_size=1, max_length=max_length, f0_threshold=0,
                                  upsampling_factor=conf.upsampling_factor, shuffle=False, epochs=1)
    sd = torch.load(ckpt, map_location="cpu")["model"]
    flat = np.concatenate([v.numpy().ravel() for v in sd.values()]).astype(np.float32)
    losses = []
    for bx, bh, bt, bd, bb in gen:
        x, h, t, d = (np.asarray(a) for a in (bx, bh, bt, bd))
        lg, _ = TO.forward(TINY, flat, x, h.astype(np.float32), d.astype(np.float32), np.asarray(bb))
        BL = int(np.asarray(bb)[0])
        losses.append(TO.ce_loss(lg, t[:, -BL:])[0])
    assert losses
    return float(np.mean(losses))


def _small_geo(fmt="npy"):
    return ["--n_resch", "32", "--n_skipch", "32", "--dilationF_depth", "2", "--dilationF_repeat", "1", "--dilationA_depth", "1",
            "--dilationA_repeat", "1", "--feature_format", fmt, "--batch_length", "1500", "--max_length", "4000", "--verbose", "0"]


def test_run_train_two_ranks(cuda, tmp_path, monkeypatch):
    """python -m qpnet_amd.run_train --n_gpus 2 end to end: the entry point re-launches itself as two ranks under torch.distributed.run (here both
    on GPU 0 over gloo: QPN_BENCH_ONE_GPU / QPN_DIST_BACKEND, RCCL refuses two ranks on one device), rank r reads chunks r, r + 2, ... of the shared
    stream, the flat gradient is exchanged once per step, rank 0 alone writes checkpoints -- and before the final model is written the run itself
    checks that the replicas are still bit-identical (parallel.replica_drift; a drift makes the entry point fail)."""
    from qpnet_amd import runners
    monkeypatch.setenv("QPN_BENCH_ONE_GPU", "1"); monkeypatch.setenv("QPN_DIST_BACKEND", "gloo"); monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = _corpus(str(tmp_path / "corpus"), n=4)
    exp = str(tmp_path / "exp")
    os.makedirs(exp)
    rc = runners.run_train(["--waveforms", root + "/wav", "--feats", root + "/feat", "--stats", root + "/stats.npz"] + _small_geo() +
                           ["--expdir", exp, "--config", exp + "/model.conf", "--iters", "4", "--checkpoint_interval", "2", "--intervals", "2",
                            "--resume", exp + "/none.pkl", "--n_gpus", "2", "--verbose", "1"])
    assert rc == 0
    assert sorted(f for f in os.listdir(exp) if f.startswith("checkpoint")) == ["checkpoint-2.pkl", "checkpoint-4.pkl", "checkpoint-final.pkl"]
    rec = yaml.safe_load(open(exp + "/loss-final.yml"))
    assert len(rec) == 2 and all(np.isfinite(rec))


def test_run_train_does_not_write_a_flagged_model(cuda, tmp_path, monkeypatch):
    """in run_train's default (lagged) mode the device status of the last steps used to be outstanding when the final model was saved.
    A target outside [0, n_quantize) injected into the LAST step of a run must raise out of run_train (the reference asserts in the step,
    qpnet_train.py:525) and leave no checkpoint-final.pkl behind."""
    from qpnet_amd import _lib, runners
    from qpnet_amd.train import FusedTrainer
    root = _corpus(str(tmp_path / "corpus"))
    exp = str(tmp_path / "exp")
    os.makedirs(exp)
    orig = FusedTrainer.step

    def step(self, x, h, t, d, b, **kw):
        if self.step_count == 4:                    # the fifth and last step
            t = t.clone(); t[0, -3] = 999
        return orig(self, x, h, t, d, b, **kw)
    monkeypatch.setattr(FusedTrainer, "step", step)
    with pytest.raises(_lib.QpnError) as e:
        runners.run_train(["--waveforms", root + "/wav", "--feats", root + "/feat", "--stats", root + "/stats.npz"] + _small_geo() +
                          ["--expdir", exp, "--config", exp + "/model.conf", "--iters", "5", "--checkpoint_interval", "100", "--intervals", "100",
                           "--resume", exp + "/none.pkl"])
    assert e.value.code == -4
    assert not os.path.exists(exp + "/checkpoint-final.pkl")


def test_staged_chunks_arrive_on_a_copy_stream_and_are_joined_by_their_first_use(cuda, monkeypatch):
    """runners.PinnedStager copies a chunk on a stream of its own; the tensors carry {ready event, device buffer} until the step / forward that first uses them
    has made its stream wait for the copy (train.join_staged).  Same logits and the same step as with the chunk copied on the compute stream."""
    import torch
    from qpnet_amd.runners import PinnedStager
    from qpnet_amd.train import FusedTrainer
    import util
    cfg = TINY
    flat = synth.make_weights(cfg, 9)
    x, h, t, d, b = synth.train_inputs(cfg, 600, 33, 30000)
    host = {"x": torch.from_numpy(x), "h": torch.from_numpy(h), "t": torch.from_numpy(t), "d": torch.from_numpy(d)}
    outs, ws = [], []
    for mode in ("1", "0"):
        monkeypatch.setenv("QPN_STAGE_STREAM", mode)
        stage = PinnedStager(cuda)
        assert (stage.copy_stream is not None) == (mode == "1")
        m = util.build_model(cfg, flat, cuda).train()
        dv = stage(host)
        assert ("_qpn_staged" in dv["x"].__dict__) == (mode == "1")
        for k in host:
            assert dv[k].device.type == "cuda" and dv[k].dtype == host[k].dtype and tuple(dv[k].shape) == tuple(host[k].shape)
        with torch.no_grad():
            outs.append(m(dv["x"], dv["h"], dv["d"], b).cpu().numpy())          # QPNet.forward joins the copy
        assert "_qpn_staged" not in dv["x"].__dict__
        for k in host:
            np.testing.assert_array_equal(dv[k].cpu().numpy(), host[k].numpy())
        tr = FusedTrainer(m, lr=1e-3)
        for i in range(6):                                                         # the ring of pinned slots (depth 4) is reused
            dv = stage(host)
            tr.step(dv["x"], dv["h"], dv["t"], dv["d"], b, want_loss=False, maxd=int(np.ceil(d.max())))
            assert "_qpn_staged" not in dv["x"].__dict__
        tr.check_status()
        ws.append(m.flat_parameters().detach().cpu().numpy())
        if mode == "1":
            # forward_loss (what run_validate calls) joins the copy too, whichever of its inputs carries the tag, and gives the oracle's loss
            from oracle import train_oracle as TO
            w_now = ws[-1]
            dv = stage(host)
            del dv["x"].__dict__["_qpn_staged"]                                  # (a consumer that looked at x alone would miss the copy)
            v = tr.forward_loss(dv["x"], dv["h"], dv["t"], dv["d"], b, maxd=int(np.ceil(d.max())))
            assert all("_qpn_staged" not in dv[k].__dict__ for k in host)
            lg, _ = TO.forward(cfg, w_now, x, h, d, b)
            assert abs(v - TO.ce_loss(lg, t[:, -int(b[0]):])[0]) < 1e-4
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_allclose(ws[0], ws[1], rtol=0, atol=2e-6)                    # (float-atomics order inside a step)
