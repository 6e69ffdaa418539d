#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the Python reference.

Runs only in the build container (needs /root/reference); the fixtures it writes are plain
data (seeds, small inputs, expected outputs) and are what travels to the GPU box.
Inputs are regenerated from seeds by qpnet_amd.synth (np.random.RandomState: frozen streams),
so fixtures store only what cannot be regenerated: the reference's outputs.

    python tests/golden/make_golden.py [--only decode|decode2|decode_d|forward|train|kat|default|deep]
"""
import argparse
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src/nets")

import torch  # noqa: E402
import qpnet as ref  # noqa: E402  (the reference module)

from qpnet_amd import synth  # noqa: E402
from qpnet_amd.config import TINY, PAPER, QPNetConfig  # noqa: E402
sys.path.insert(0, HERE)
from cases import (DECODE_CASES, DECODE_CASES2, DECODE_CASES_D, FORWARD_CASES, TRAIN_CASES, FORWARD_CASES_D, TRAIN_CASES_D,  # noqa: E402
                   FORWARD_CASES_DEEP, TRAIN_CASES_DEEP, DECODE_CASES_DEEP, decode2_inputs)

torch.set_num_threads(8)
torch.set_grad_enabled(False)


def build_ref(cfg, flat):
    m = ref.QPNet(**cfg.kwargs())
    sd = m.state_dict()
    layout = cfg.param_layout()
    assert list(sd.keys()) == [k for k, _ in layout], "state_dict order differs from QPNetConfig.param_layout"
    for k, shp in layout:
        assert tuple(sd[k].shape) == tuple(shp), (k, sd[k].shape, shp)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
    m.eval()
    return m


def gen_kat():
    x = np.linspace(-1.0, 1.0, 4097)
    enc = ref.encode_mu_law(x, 256)
    dec = ref.decode_mu_law(np.arange(256), 256)
    rs = np.random.RandomState(3)
    out = {"mulaw_x": x, "mulaw_enc": enc, "mulaw_dec": dec}
    m = ref.QPNet(**TINY.kwargs())
    # dilated indices: f0 in [45,450]*{0.5,1,1.5}; float32 tensor path and float64 numpy path
    f0 = rs.uniform(45, 450, size=(2, 3000))
    for fac, tag in ((0.5, "h"), (1.0, "1"), (1.5, "x")):
        d64 = 22050.0 / (f0 * fac) / 8.0
        d32 = d64.astype(np.float32)
        out["didx_d64_" + tag] = d64
        for k in range(4):
            dil = 2 ** k
            out["didx_train_f32_%s_%d" % (tag, k)] = m._dilated_index(torch.from_numpy(d32), dil, 1)[:, 0].numpy()
            out["didx_train_f64_%s_%d" % (tag, k)] = m._dilated_index(d64, dil, 1, tensor=False)[:, 0]
            out["didx_gen_f32_%s_%d" % (tag, k)] = m._generate_dilated_index(torch.from_numpy(d32), dil, 1)[:, 0].numpy()
            out["didx_gen_f64_%s_%d" % (tag, k)] = m._generate_dilated_index(d64, dil, 1, tensor=False)[:, 0]
    # a long one near the float32 rounding hazard (|idx| > 16384)
    dl = (22050.0 / rs.uniform(45, 450, size=(1, 21000)) / 8.0).astype(np.float32)
    out["didx_long_d32"] = dl
    out["didx_long_train_f32_3"] = m._dilated_index(torch.from_numpy(dl), 8, 1)[:, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **out)
    print("kat.npz written")




def gen_decode(cases=DECODE_CASES, fname="decode.npz"):
    from qpnet_amd import harness
    out = {}
    for name, cfg, wseed, utts, extra in cases:
        flat = synth.make_weights(cfg, wseed)
        m = build_ref(cfg, flat)
        xs, hs, ds, ns = [], [], [], []
        for (fs, nf, fac) in utts:
            x, h, d, n = synth.decode_inputs(cfg, nf, fs, fac)
            xs.append(x); hs.append(h.T); ds.append(d[:, None]); ns.append(n)
        bx = torch.from_numpy(np.stack(xs)).long()
        bh = torch.from_numpy(harness.pad_list(hs)).float().transpose(1, 2)
        bd = harness.pad_list(ds).squeeze(-1)
        if extra:
            bd = torch.from_numpy(bd).float()
        nlist = list(ns)
        streams = m.batch_fast_generate(bx, bh, nlist, bd, intervals=None, mode="argmax", extra_memory=extra)
        for i, s in enumerate(streams):
            out["%s_out%d" % (name, i)] = np.asarray(s).astype(np.int16)
        out[name + "_nleft"] = np.array(nlist, dtype=np.int64)
        print(name, [len(s) for s in streams], "n_samples_list after:", nlist)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "written")




def gen_decode2():
    out = {}
    for case in DECODE_CASES2:
        cfg, name = case["cfg"], case["name"]
        m = build_ref(cfg, synth.make_weights(cfg, case["wseed"]))
        bx, bh, bd, ns = decode2_inputs(case)
        bd_ = torch.from_numpy(bd).float() if case["extra"] else bd
        nlist = list(ns)
        streams = m.batch_fast_generate(torch.from_numpy(bx).long(), torch.from_numpy(bh).float(), nlist, bd_, intervals=None,
                                        mode="argmax", extra_memory=case["extra"])
        for i, s in enumerate(streams):
            out["%s_out%d" % (name, i)] = np.asarray(s).astype(np.int16)
        out[name + "_nleft"] = np.array(nlist, dtype=np.int64)
        out[name + "_maxd"] = np.int64(np.nanmax(np.ceil(bd)))
        print(name, [len(s) for s in streams], "maxd", out[name + "_maxd"], "n_samples_list after:", nlist)
    np.savez_compressed(os.path.join(HERE, "decode2.npz"), **out)
    print("decode2.npz written")


def gen_forward(cases=FORWARD_CASES, fname="forward.npz"):
    out = {}
    for name, cfg, wseed, dseed, bl, ml in cases:
        flat = synth.make_weights(cfg, wseed)
        m = build_ref(cfg, flat)
        x, h, t, d, b = synth.train_inputs(cfg, bl, dseed, ml)
        logits = m(torch.from_numpy(x), torch.from_numpy(h), torch.from_numpy(d), torch.from_numpy(b))
        BL = int(b[0])
        tgt = torch.from_numpy(t[:, -BL:]).reshape(-1)
        loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tgt)
        out[name + "_logits"] = logits.numpy().astype(np.float32)
        out[name + "_loss"] = np.float64(loss.item())
        out[name + "_bl"] = np.int64(BL)
        print(name, "forward", logits.shape, "loss", loss.item())
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "written")




def gen_train(cases=TRAIN_CASES, fname="train.npz", max_length=30000):
    """A few real optimisation steps: CE(mean) -> backward -> Adam(lr 1e-4)
    (reference src/bin/qpnet_train.py:426-430,517-531)."""
    torch.set_grad_enabled(True)
    out = {}
    for name, cfg, wseed, dseed, bl, nsteps in cases:
        flat = synth.make_weights(cfg, wseed)
        m = build_ref(cfg, flat)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=0.0)
        crit = torch.nn.CrossEntropyLoss()
        losses = []
        for step in range(nsteps):
            x, h, t, d, b = synth.train_inputs(cfg, bl, dseed + step, max_length)
            BL = int(b[0])
            logits = m(torch.from_numpy(x), torch.from_numpy(h), torch.from_numpy(d), torch.from_numpy(b))
            loss = crit(logits.reshape(-1, cfg.n_quantize), torch.from_numpy(t[:, -BL:]).reshape(-1))
            opt.zero_grad()
            loss.backward()
            if step == 0:
                g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in m.parameters()]).numpy()
                if g.size > 60000:   # keep the fixture small: per-tensor L2 norms + a strided sample
                    out[name + "_grad0_sample"] = g[::97].astype(np.float32)
                else:
                    out[name + "_grad0"] = g.astype(np.float32)
                out[name + "_grad0_norms"] = np.array([0.0 if p.grad is None else p.grad.norm().item() for p in m.parameters()])
            opt.step()
            losses.append(loss.item())
        w = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy()
        out[name + "_losses"] = np.array(losses)
        out[name + "_wfinal_sample"] = w[::97].astype(np.float32)
        print(name, "train losses", losses)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    torch.set_grad_enabled(False)
    print(fname, "written")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    todo = [a.only] if a.only else ["kat", "decode", "decode2", "decode_d", "forward", "train", "default", "deep"]
    for t in todo:
        {"kat": gen_kat, "decode": gen_decode, "decode2": gen_decode2, "forward": gen_forward, "train": gen_train,
         "decode_d": lambda: gen_decode(DECODE_CASES_D, "decode_d.npz"),
         "default": lambda: (gen_forward(FORWARD_CASES_D, "forward_d.npz"), gen_train(TRAIN_CASES_D, "train_d.npz", 2000)),
         "deep": lambda: (gen_forward(FORWARD_CASES_DEEP, "forward_deep.npz"), gen_train(TRAIN_CASES_DEEP, "train_deep.npz", 22500),
                          gen_decode(DECODE_CASES_DEEP, "decode_deep.npz"))}[t]()
