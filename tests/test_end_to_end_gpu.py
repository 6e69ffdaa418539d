"""GPU: the hot path stitched together the way run_QP.sh stages 1-3 use it -- train_generator chunks -> fused training
steps -> checkpoint -> fresh model -> decode_generator batches -> batch_fast_generate -> 16-bit wav -- with the decode
of the TRAINED weights checked bit-exactly against the CPU oracle."""
import numpy as np
import pytest

from qpnet_amd import loaders, synth
from qpnet_amd.config import TINY

pytestmark = pytest.mark.gpu


def test_train_checkpoint_decode_roundtrip(cuda, oracle, tmp_path):
    import torch
    from qpnet_amd.qpnet import QPNet, initialize
    from qpnet_amd.train import FusedTrainer
    cfg = TINY
    U = cfg.upsampling_factor
    torch.manual_seed(1)
    m = QPNet(**cfg.kwargs()); m.apply(initialize); m = m.to(cuda).train()
    rs = np.random.RandomState(3)
    utts = [(rs.uniform(-0.9, 0.9, 40 * U + 5).astype(np.float32), synth.make_features(40, 300 + i)) for i in range(2)]
    mean, scale = synth.scaler_stats()
    gen = loaders.train_generator(utts, m.receptiveCausal_field, m.receptiveF_field, m.receptiveA_field, 22050,
                                  wav_transform=loaders.mu_law_transform(256), feat_transform=lambda h: (h - mean) / scale,
                                  batch_length=1200, max_length=4000, upsampling_factor=U, shuffle=False, epochs=1, device=cuda)
    tr = FusedTrainer(m, lr=1e-3)
    losses = []
    for x, h, t, d, b in gen:
        losses.append(tr.step(x, h, t, d, b))
        if len(losses) == 4:
            break
    assert len(losses) >= 3 and all(np.isfinite(losses)) and losses[0] < 8.0
    path = loaders.save_checkpoint(str(tmp_path), m, None, len(losses))
    m2 = QPNet(**cfg.kwargs())
    assert loaders.load_checkpoint(path, m2) == len(losses)
    m2 = m2.to(cuda).eval()
    flat = np.concatenate([v.detach().cpu().numpy().ravel() for v in m2.state_dict().values()]).astype(np.float32)
    feats = [synth.make_features(n, 400 + i) for i, n in enumerate([4, 7, 4])]
    outs = {}
    for ids, bx, bh, ns, bd in loaders.decode_generator(feats, 22050, wav_transform=loaders.mu_law_transform(256),
                                                        feat_transform=lambda h: (h - mean) / scale, batch_size=2,
                                                        upsampling_factor=U, f0_factor=1.0, device=cuda):
        want = list(ns)
        ys = m2.batch_fast_generate(bx, bh, ns, bd, mode="argmax")
        assert len(ys) == len(ids)
        # completion order == ascending length (stable), which is the generator's order inside a batch
        for fid, y, n in zip(ids, ys, want):
            assert len(y) == n
            outs[fid] = y
            loaders.write_wav(str(tmp_path / ("%s.wav" % fid)), 22050, y)
    assert sorted(outs) == ["utt0000", "utt0001", "utt0002"]
    # oracle decode with the trained weights, prepared exactly like the generator does
    for i, f in enumerate(feats):
        h = np.array(f, copy=True)
        from qpnet_amd import harness
        d = harness.extend_time(harness.dilated_factor(harness.batch_f0(h), 22050, 8)[:, None], U)[:, 0]
        hn = ((h - mean) / scale).astype(np.float32)
        ref = oracle.decode(cfg, flat, np.ascontiguousarray(hn.T), d, np.array([128], dtype=np.int64), h.shape[0] * U - 1)["samples"]
        np.testing.assert_array_equal(outs["utt%04d" % i], ref)
