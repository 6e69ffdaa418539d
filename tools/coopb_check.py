"""Dev check of the utterance-batched cooperative decode kernel (decode_coopb.hip) on the repo-default geometry:
its streams against those of the per-utterance cooperative kernel (decode_coop.hip, QPN_DECODE_COOPB=0 -- itself pinned
to the oracle and the reference's golden streams by tests/test_decode_gpu.py), ragged batches, both modes; then timing.

    python tools/coopb_check.py [--frames 40] [--time-frames 200]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(cfg, flat, dev, coopb):
    import torch
    from qpnet_amd import synth
    from qpnet_amd.qpnet import QPNet
    if coopb is None:
        os.environ.pop("QPN_DECODE_COOPB", None)
    else:
        os.environ["QPN_DECODE_COOPB"] = str(coopb)
    m = QPNet(**cfg.kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
    return m.to(dev).eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--time-frames", type=int, default=200)
    ap.add_argument("--batches", default="4,5,20,37")
    ap.add_argument("--time-batches", dest="time_batches", default="4,8,16,20,32,64")
    args = ap.parse_args()
    import torch
    from qpnet_amd import synth
    from qpnet_amd.config import DEFAULT as cfg
    dev = torch.device("cuda", 0)
    flat = synth.make_weights(cfg, 7)
    bad = 0
    for B in [int(b) for b in args.batches.split(",")]:
        utts = [(300 + b, max(1, args.frames - (b % 4)), 0.5 + 0.25 * (b % 5)) for b in range(B)]
        bx, bh, bd, ns = synth.decode_batch(cfg, utts)
        xb, hb = torch.from_numpy(bx).to(dev), torch.from_numpy(bh).to(dev)
        for mode in ("argmax", "sampling"):
            res = {}
            for name, knob in (("coopb", 1), ("coop", 0)):
                m = build(cfg, flat, dev, knob)
                m.sampling_seed = 5
                outs = m.batch_fast_generate(xb, hb, list(ns), bd, mode=mode)
                res[name] = (outs, m.last_decode_plan, m.last_decode_kernel_ms)
                del m
            same = all(np.array_equal(a, b) for a, b in zip(res["coopb"][0], res["coop"][0]))
            bad += 0 if same else 1
            first = ""
            if not same:
                for i, (a, b) in enumerate(zip(res["coopb"][0], res["coop"][0])):
                    if not np.array_equal(a, b):
                        j = int(np.argmax(a != b)); first = " first diff: row %d (len %d) sample %d: %d vs %d" % (i, len(a), j, a[j], b[j]); break
            print("B=%d %s: %s  [%s | %.2f ms] vs [%s | %.2f ms]%s" % (B, mode, "same" if same else "DIFFERENT", res["coopb"][1], res["coopb"][2],
                                                                 res["coop"][1], res["coop"][2], first), flush=True)
    # timing: equal-length utterances
    for B in [int(b) for b in args.time_batches.split(",")]:
        utts = [(100 + b, args.time_frames, 1.0) for b in range(B)]
        bx, bh, bd, ns = synth.decode_batch(cfg, utts)
        xb, hb = torch.from_numpy(bx).to(dev), torch.from_numpy(bh).to(dev)
        line = "B=%d x %d samples:" % (B, ns[0])
        for name, knob in (("coopb", 1), ("coop", 0)):
            m = build(cfg, flat, dev, knob)
            m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            line += "  %s %.1f k samples/s (%.1f us/step, %s)" % (name, sum(ns) / dt / 1e3, m.last_decode_kernel_ms * 1e3 / max(ns), m.last_decode_plan)
            del m
        print(line, flush=True)
    print("COOPB_CHECK", "OK" if bad == 0 else "FAILED %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
