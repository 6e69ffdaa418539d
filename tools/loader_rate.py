"""How many training chunks per second does the host side deliver?  (What's weak 2)

Times loaders.train_generator -- the counterpart of the reference's generator (src/bin/qpnet_train.py:200-335) -- on synthetic
utterances of VCC2018 shape (3-6 s at 22.05 kHz, 5 ms frames), paper-size receptive fields, batch_length 20000 /
max_length 30000 (src/utils/param_model.py:63), with the mu-law and scaler transforms the trainer passes, for
shard = None and for rank 0 of 2 / 4 / 8 ranks.  With a GPU it also times the per-chunk host-to-device copies.
The fused step consumes ~1.07 k chunks/s per GPU (BENCH_r02); compare.

    python tools/loader_rate.py [--chunks 200]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpnet_amd import loaders, synth          # noqa: E402
from qpnet_amd.config import PAPER            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=200)
    ap.add_argument("--files", action="store_true", help="file-backed corpus (int16 .wav + .npy features in a temporary directory) through runners._utterance_loaders")
    args = ap.parse_args()
    import torch
    cfg = PAPER
    U = cfg.upsampling_factor
    rs = np.random.RandomState(0)
    utts = []
    for i in range(24):
        nf = int(rs.randint(600, 1200))
        utts.append((rs.uniform(-1, 1, nf * U + 5).astype(np.float32), synth.make_features(nf, 400 + i)))
    if args.files:
        import tempfile
        from scipy.io import wavfile
        from qpnet_amd import runners
        root = tempfile.mkdtemp(prefix="qpn_loader_rate_")
        wavs, feats = [], []
        for i, (x, h) in enumerate(utts):
            wavs.append("%s/u%03d.wav" % (root, i)); feats.append("%s/u%03d.npy" % (root, i))
            wavfile.write(wavs[-1], 22050, (x * 32767).astype(np.int16)); np.save(feats[-1], h)
        utts = runners._utterance_loaders(wavs, feats, "world")
    mean, scale = synth.scaler_stats()
    scaler = lambda h: (h - mean) / scale      # noqa: E731
    dev = torch.device("cuda:0") if torch.cuda.is_available() else None

    from qpnet_amd.runners import PinnedStager
    stage = PinnedStager(dev) if dev is not None else None

    def rate(shard, to_dev):
        np.random.seed(1)
        gen = loaders.train_generator(utts, cfg.receptiveCausal_field, cfg.receptiveF_field, cfg.receptiveA_field, 22050,
                                      wav_transform=loaders.mu_law_transform(256), feat_transform=scaler, batch_length=20000,
                                      max_length=30000, upsampling_factor=U, shuffle=True, shard=shard)
        next(gen)
        t0 = time.time()
        for _ in range(args.chunks):
            bx, bh, bt, bd, bb = next(gen)
            maxd = int(np.ceil(float(bd.max())))          # what runners._batches does per chunk
            if to_dev == "pinned":
                out = stage({"x": bx, "h": bh, "t": bt, "d": bd})
            elif to_dev:
                out = [t.to(dev, non_blocking=True) for t in (bx, bh, bt, bd)]
        if to_dev:
            torch.cuda.synchronize()
        r = args.chunks / (time.time() - t0)
        rate.loads = loaders.train_generator.last_stats["loads"]
        return r

    print("corpus: %s" % ("24 utterances as .wav + .npy files (runners._FileUtterance: plan() = wav header + features)" if args.files else "24 utterances in memory"))
    print("train_generator, paper-size, batch_length 20000 (one host thread):")
    for shard in (None, (0, 2), (0, 4), (0, 8)):
        r = rate(shard, False)
        print("  shard %-8s %8.1f chunks/s delivered to this rank (host only); %d full utterance loads for %d chunks" % (shard, r, rate.loads, args.chunks + 1))
    if dev is not None:
        print("  shard None     %8.1f chunks/s incl. host-to-device copies from pageable memory (tensor.to)" % rate(None, True))
        print("  shard (0, 8)   %8.1f chunks/s incl. host-to-device copies from pageable memory" % rate((0, 8), True))
        print("  shard None     %8.1f chunks/s incl. host-to-device copies through the pinned staging ring (runners.PinnedStager)" % rate(None, "pinned"))
        print("  shard (0, 8)   %8.1f chunks/s incl. host-to-device copies through the pinned staging ring" % rate((0, 8), "pinned"))


if __name__ == "__main__":
    main()
