"""north_star: "launches once per utterance under hipGraph" -- the decode side of that sentence, measured.

A decode call is ONE persistent kernel plus four small pre-pass nodes (descriptor copy, two memsets, k_known, k_aux_project).  This tool captures exactly that
enqueue (qpn_decode_enqueue on a capturing stream) into a hipGraph and replays it, against the eager enqueue, for B = 1 and a 10-frame utterance (1 099 samples:
the shortest call the reference makes, where launch overhead weighs most) and for a 200-frame one; the replayed stream must equal the eager one sample for sample.
    python tools/graph_decode.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
from qpnet_amd import _lib, synth              # noqa: E402
from qpnet_amd.config import PAPER             # noqa: E402
from qpnet_amd.qpnet import QPNet              # noqa: E402

dev = torch.device("cuda:0")
cfg = PAPER
flat = synth.make_weights(cfg, 13)
m = QPNet(**cfg.kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
m = m.to(dev).eval()
L, hd = m._native(dev)

for frames in (10, 200):
    x, h, d, ns = synth.decode_batch(cfg, [(100, frames, 1.0)])
    xd, hh = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    dd = torch.from_numpy(np.ascontiguousarray(d, dtype=np.float64)).to(dev)
    maxd = int(np.nanmax(np.ceil(d)))
    ref = m.batch_fast_generate(xd, hh, list(ns), d, mode="argmax")[0]          # eager, through the module (also binds the weights)
    out = torch.empty((1, ns[0]), dtype=torch.int64, device=dev)
    arr = (C.c_int64 * 1)(*ns)

    def enqueue(stream):
        _lib.check(L.qpn_decode_enqueue(hd, 1, xd.shape[1], hh.shape[2], dd.shape[1], xd.data_ptr(), hh.data_ptr(), dd.data_ptr(), 0, arr, maxd, 0, 0,
                                        None, out.data_ptr(), None, stream))

    def eager_once():
        s = torch.cuda.current_stream(dev).cuda_stream
        enqueue(s)
        _lib.check(L.qpn_decode_finish(hd, s))

    for _ in range(3):
        eager_once()
    assert np.array_equal(out[0].cpu().numpy(), ref)
    N = 30
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        eager_once()
    t_eager = (time.perf_counter() - t0) / N
    k_ms = float(L.qpn_last_decode_kernel_ms(hd))
    # capture
    g = torch.cuda.CUDAGraph()
    out.zero_()
    try:
        with torch.cuda.graph(g):
            enqueue(torch.cuda.current_stream(dev).cuda_stream)
        captured = True
    except Exception as e:                      # (a call the runtime refuses while capturing)
        captured = False
        print("frames %d: capture failed: %r" % (frames, e))
    if captured:
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        same = np.array_equal(out[0].cpu().numpy(), ref)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N):
            g.replay()
            torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / N
        print("B = 1, %d frames (%d samples): eager enqueue + finish %.3f ms per call (persistent kernel %.3f ms), hipGraph replay + synchronise %.3f ms per call; "
              "replayed stream == eager stream: %s" % (frames, ns[0], t_eager * 1e3, k_ms, t_graph * 1e3, same))
    # leave the handle consistent for the next round (the captured enqueue marked a decode as pending)
    try:
        L.qpn_decode_finish(hd, torch.cuda.current_stream(dev).cuda_stream)
    except Exception:
        pass
