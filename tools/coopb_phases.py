"""Dev: per-phase times of k_decode_coopb (a -DQPN_ENABLE_STAMPS build: python qpnet_amd/csrc/build.py --variant stamps -DQPN_ENABLE_STAMPS;
QPN_LIB=build_variants/libqpnet_stamps.so QPN_STAMPS=1 python tools/coopb_phases.py [B] [frames]).  The library prints the stamps on stderr:
stamp k (k = 1..11) = cycles summed over the steps in phase k - 1 (P1, P2 compute, P2 gather, S1, S2 publish, S2 gather, S3, S4 publish, S4 gather, tail, pick)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("QPN_DECODE_COOPB", "1")      # (the batched kernel at every batch size)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import torch
    from qpnet_amd import synth
    from qpnet_amd.config import DEFAULT as cfg
    from qpnet_amd.qpnet import QPNet
    dev = torch.device("cuda", 0)
    flat = synth.make_weights(cfg, 7)
    m = QPNet(**cfg.kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
    m = m.to(dev).eval()
    bx, bh, bd, ns = synth.decode_batch(cfg, [(100 + b, frames, 1.0) for b in range(B)])
    xb, hb = torch.from_numpy(bx).to(dev), torch.from_numpy(bh).to(dev)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("B=%d x %d samples: %.1f k samples/s, kernel %.2f ms = %.1f us/step, plan %s" % (B, ns[0], sum(ns) / dt / 1e3, m.last_decode_kernel_ms,
                                                                                 m.last_decode_kernel_ms * 1e3 / max(ns), m.last_decode_plan), flush=True)


if __name__ == "__main__":
    main()
