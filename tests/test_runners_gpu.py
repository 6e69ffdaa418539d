"""GPU: the task entry points (python -m qpnet_amd.run_train|run_update|run_validate|run_decode) on a small synthetic
corpus of .wav + .npy feature files: stages 1-3 of run_QP.sh on the native path, incl. checkpoint resume."""
import argparse
import os

import numpy as np
import pytest
import yaml

from qpnet_amd import loaders, synth
from qpnet_amd.config import TINY

pytestmark = pytest.mark.gpu


def _corpus(root, n=3, frames=45):
    from scipy.io import wavfile
    U = TINY.upsampling_factor
    os.makedirs(root + "/wav"); os.makedirs(root + "/feat")
    rs = np.random.RandomState(5)
    feats = []
    for i in range(n):
        h = synth.make_features(frames + 3 * i, 700 + i)
        x = (rs.uniform(-0.8, 0.8, (frames + 3 * i) * U + 11) * 32767).astype(np.int16)
        wavfile.write("%s/wav/u%02d.wav" % (root, i), 22050, x)
        np.save("%s/feat/u%02d.npy" % (root, i), h)
        feats.append(h)
    st = loaders.calc_stats(feats)
    np.savez(root + "/stats.npz", mean=st.mean_, scale=st.scale_)
    return root


def test_train_update_validate_decode_entry_points(cuda, tmp_path, oracle):
    import torch
    from qpnet_amd import runners
    root = _corpus(str(tmp_path / "corpus"))
    exp = str(tmp_path / "exp")
    common = ["--waveforms", root + "/wav", "--feats", root + "/feat", "--stats", root + "/stats.npz"]
    geo = ["--n_resch", "32", "--n_skipch", "32", "--dilationF_depth", "2", "--dilationF_repeat", "1", "--dilationA_depth", "1",
           "--dilationA_repeat", "1", "--feature_format", "npy", "--batch_length", "1500", "--max_length", "4000", "--verbose", "0"]
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
    # decode (greedy so the oracle can check it), wav files named by feature id
    out = str(tmp_path / "wav_out")
    assert runners.run_decode(["--feats", root + "/feat", "--stats", root + "/stats.npz", "--config", conf, "--checkpoint",
                               exp + "/checkpoint-final.pkl", "--outdir", out + "/feat_id.wav", "--batch_size", "2", "--mode", "argmax",
                               "--intervals", "2000", "--verbose", "0"]) == 0
    from scipy.io import wavfile
    sd = torch.load(exp + "/checkpoint-final.pkl", map_location="cpu")["model"]
    flat = np.concatenate([v.numpy().ravel() for v in sd.values()]).astype(np.float32)
    sc = loaders.read_scaler_stats(root + "/stats.npz")
    for i in range(3):
        fs, w = wavfile.read("%s/u%02d.wav" % (out, i))
        h = np.load("%s/feat/u%02d.npy" % (root, i))
        assert fs == 22050 and w.dtype == np.int16 and len(w) == h.shape[0] * TINY.upsampling_factor - 1
    # the shortest utterance re-decoded by the oracle from the checkpoint == the wav that was written
    h = np.load(root + "/feat/u00.npy")
    from qpnet_amd import harness
    d = harness.extend_time(harness.dilated_factor(harness.batch_f0(h), 22050, 8)[:, None], TINY.upsampling_factor)[:, 0]
    # batch-level maxd: u00 was decoded together with u01 (batch_size 2, sorted by length)
    h1 = np.load(root + "/feat/u01.npy")
    d1 = harness.dilated_factor(harness.batch_f0(h1), 22050, 8)
    maxd = int(np.ceil(max(d.max(), d1.max())))
    hn = sc.transform(h).astype(np.float32)
    ref = oracle.decode(TINY, flat, np.ascontiguousarray(hn.T), d, np.array([128], dtype=np.int64), h.shape[0] * 110 - 1, maxd=maxd)["samples"]
    np.testing.assert_array_equal(wavfile.read(out + "/u00.wav")[1], loaders.samples_to_int16(ref))
