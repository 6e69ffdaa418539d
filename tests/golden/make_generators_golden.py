#!/usr/bin/env python3
"""Golden data for the batch generators either side of the hot path, produced by RUNNING the reference's own
generators in the build container (needs /root/reference; the output generators.npz is plain data and travels):

  train_generator   /root/reference/src/bin/qpnet_train.py:200-335
  decode_generator  /root/reference/src/bin/qpnet_decode.py:122-209

The two task scripts import h5py and torchvision, which this image lacks; empty stand-in modules are registered for
the import (SURVEY.md §8c did the same for the pure helpers) and the scripts' `read_hdf5` / `shape_hdf5` names are
pointed at an in-memory table of synthetic WORLD-shaped features.  Waveforms are REAL 16-bit wav files written with
scipy into a temporary directory (the reference reads them with scipy.io.wavfile).  Nothing else is replaced: the
generator bodies, `_validate_length`, `_dilated_factor`, `extend_time`, `pad_list`, the background-thread wrapper and
numpy's global RNG for the shuffles run as they are.

Stored per yielded batch: shapes, batch_length_current, CRC32 of the bytes of x / h / t / d (inputs are regenerated
from seeds by `generator_corpus` below, shared with tests/test_loaders_cpu.py).

    python tests/golden/make_generators_golden.py
"""
import os
import sys
import tempfile
import types
import warnings
import zlib

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from cases import GEN_TRAIN_CASES, GEN_DECODE_CASES, generator_corpus, crc  # noqa: E402


def import_reference_scripts():
    for name in ("h5py", "torchvision"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tv.transforms = types.ModuleType("torchvision.transforms")
        sys.modules["torchvision.transforms"] = tv.transforms
    sys.path.insert(0, "/root/reference/src/utils")
    sys.path.insert(0, "/root/reference/src/nets")
    sys.path.insert(0, "/root/reference/src/bin")
    import qpnet_train as ref_train
    import qpnet_decode as ref_decode
    return ref_train, ref_decode


def main():
    import torch
    from scipy.io import wavfile
    from sklearn.preprocessing import StandardScaler
    ref_train, ref_decode = import_reference_scripts()
    import qpnet as ref_qpnet
    out = {}
    tmp = tempfile.mkdtemp(prefix="qpn_gen_")
    table = {}

    def fake_read_hdf5(path, key):
        assert key == "/world"
        return table[os.path.basename(path)].copy()

    def fake_shape_hdf5(path, key):
        assert key == "/world"
        return table[os.path.basename(path)].shape
    ref_train.read_hdf5 = fake_read_hdf5
    ref_decode.read_hdf5 = fake_read_hdf5
    ref_decode.shape_hdf5 = fake_shape_hdf5

    def scaler_of(mean, scale):
        sc = StandardScaler()
        sc.mean_, sc.scale_ = mean, scale
        return sc.transform

    # ------------------------------------------------------------------ train_generator
    for case in GEN_TRAIN_CASES:
        name = case["name"]
        pcm, feats, mean, scale = generator_corpus(case["corpus_seed"], case["frames"], case["sample_slack"], case["f0"], case["U"])
        wavs, h5s = [], []
        for i, (p, f) in enumerate(zip(pcm, feats)):
            w = os.path.join(tmp, "%s_u%d.wav" % (name, i)); h = os.path.join(tmp, "%s_u%d.h5" % (name, i))
            wavfile.write(w, 22050, p)
            table[os.path.basename(h)] = f
            wavs.append(w); h5s.append(h)
        np.random.seed(case["np_seed"])
        gen = ref_train.train_generator(
            wavs, h5s, case["rc"], case["rf"], case["ra"], wav_transform=lambda x: ref_qpnet.encode_mu_law(x, 256),
            feat_transform=scaler_of(mean, scale), feature_type="world", dense_factor=8, batch_length=case["batch_length"],
            batch_size=case["batch_size"], max_length=case["max_length"], f0_threshold=case["f0_threshold"],
            upsampling_factor=case["U"], shuffle=case["shuffle"])
        rows = []
        for _ in range(case["n_batches"]):
            bx, bh, bt, bd, bb = next(gen)
            assert bx.dtype == torch.int64 and bh.dtype == torch.float32 and bd.dtype == torch.float32
            rows.append([bx.shape[0], bx.shape[1], bh.shape[2], int(bb[0]), crc(bx.numpy()), crc(bh.numpy()), crc(bt.numpy()),
                         crc(bd.numpy()), int(bb.sum())])
        out["train_" + name] = np.array(rows, dtype=np.int64)
        print(name, "batches", len(rows), "T/BL of the first five:", [(r[1], r[3]) for r in rows[:5]],
              "distinct BL:", sorted(set(r[3] for r in rows)))

    # ------------------------------------------------------------------ decode_generator
    for case in GEN_DECODE_CASES:
        name = case["name"]
        _, feats, mean, scale = generator_corpus(case["corpus_seed"], case["frames"], None, case["f0"], case["U"])
        h5s = []
        for i, f in enumerate(feats):
            h = os.path.join(tmp, "%s_d%d.h5" % (name, i))
            table[os.path.basename(h)] = f
            h5s.append(h)
        gen = ref_decode.decode_generator(
            h5s, 22050, wav_transform=lambda x: ref_qpnet.encode_mu_law(x, 256), feat_transform=scaler_of(mean, scale),
            feature_type="world", feat_ext=".h5", dense_factor=8, batch_size=case["batch_size"], upsampling_factor=case["U"],
            f0_factor=float(str(case["f0_factor"])), f0_dim_index=1, extra_memory=case["extra_memory"])
        rows, ids_all, ns_all = [], [], []
        for feat_ids, bx, bh, ns, bd in gen:
            bdn = bd.numpy() if case["extra_memory"] else bd
            assert bdn.dtype == (np.float32 if case["extra_memory"] else np.float64)
            rows.append([len(feat_ids), bx.shape[1], bh.shape[2], bdn.shape[1], crc(bx.numpy()), crc(bh.numpy()), crc(bdn)])
            ids_all += [int(s.split("_d")[1]) for s in feat_ids]
            ns_all += list(ns)
        out["decode_" + name] = np.array(rows, dtype=np.int64)
        out["decode_" + name + "_order"] = np.array(ids_all, dtype=np.int64)
        out["decode_" + name + "_ns"] = np.array(ns_all, dtype=np.int64)
        print(name, "batches", [r[0] for r in rows], "order", ids_all, "n_samples", ns_all)
    np.savez_compressed(os.path.join(HERE, "generators.npz"), **out)
    print("generators.npz written")


if __name__ == "__main__":
    main()
