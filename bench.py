#!/usr/bin/env python3
"""bench.py -- QPNet hot-path benchmark on MI355X (contract: see the task prompt / DESIGN.md §6).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode decode|train] [--batch B]

One "step" is one pass of the hot path over one batch of synthetic WORLD-shaped input:
  decode: QPNet.batch_fast_generate(mode="argmax") of B 10-second utterances @22.05 kHz
          (paper-size QPNet, F=2005 frames -> 220 549 samples each), one persistent kernel launch;
  train : one optimisation step (forward + CE + backward + Adam) on one chunk of RF+20000 samples.
Rank 0 prints ONE JSON line.  Multi-GPU: one process per GPU (torch.distributed / RCCL); decode
runs independent replicas (no collective, "replicas only"), training all-reduces gradients.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak


def dist_setup(n_gpus):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return rank, local, world


def barrier(world):
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(v, world, dev):
    import torch
    import torch.distributed as dist
    if world == 1:
        return v
    t = torch.tensor([v], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def decode_weight_bytes_per_sample(cfg):
    """172 B of per-sample inputs/outputs + 4 B x the parameters touched per sample
    (all but the one-hot table, of which two columns are looked up)."""
    C, Q = cfg.n_resch, cfg.n_quantize
    touched = cfg.n_params - Q * C * 2 + 2 * C - (cfg.upsampling_factor + 1)
    return 172 + 4 * touched


def cpu_baseline_decode(cfg, flat, n_frames=600):
    """The CPU oracle (a single-threaded C port of the reference algorithm, validated against
    the reference's own streams) timed on this box's host cores on a bounded sample."""
    from oracle import cpu_oracle
    from qpnet_amd import synth
    x, h, d, n = synth.decode_inputs(cfg, n_frames, 1, 1.0)
    cpu_oracle.lib()
    t0 = time.time()
    cpu_oracle.decode(cfg, flat, h, d, x, n)
    dt = time.time() - t0
    return {"value": n / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "greedy decode of one %d-frame utterance (%d samples), same synthetic features/weights" % (n_frames, n)}


def run_decode(args, rank, local, world):
    import torch
    from qpnet_amd import synth
    from qpnet_amd.config import PAPER
    from qpnet_amd.qpnet import QPNet
    dev = torch.device("cuda", local)
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    m = QPNet(**cfg.kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
    m = m.to(dev).eval()
    B, F = args.batch, args.frames
    utts = [(1000 * rank + 100 + b, F, 1.0) for b in range(B)]
    bx, bh, bd, ns = synth.decode_batch(cfg, utts)
    xb, hb = torch.from_numpy(bx).to(dev), torch.from_numpy(bh).to(dev)

    def step():
        return m.batch_fast_generate(xb, hb, list(ns), bd, mode="argmax")

    for _ in range(args.warmup):
        step()
    kms = []
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kms.append(m.last_decode_kernel_ms)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    total_samples = sum(ns) * args.steps * world
    value = total_samples / dt
    k_ms = float(np.mean(kms))
    bps = decode_weight_bytes_per_sample(cfg)
    achieved = bps * sum(ns) / (k_ms * 1e-3) / 1e9
    out = {
        "metric": "AR decode samples/sec/GPU @22.05kHz (greedy)" if world == 1 else "AR decode samples/sec @22.05kHz (greedy), all GPUs",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config[3]: batch_fast_generate(argmax) of %d x %.1f s utterances per GPU, paper-size QPNet "
                               "(C=64,S=256,4F+4A), F=%d frames -> %d samples each" % (B, ns[0] / 22050.0, F, ns[0]),
                   "batch_per_gpu": B, "parallelism": "replicas x%d (no collective)" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_decode", "kernel_ms": k_ms,
                     "note": "algorithmic bytes = (172 + 4*params touched) B/sample = %d B/sample (SURVEY 8d, weights re-streamed "
                             "every sample; they are L2-resident so this is L2->CU traffic, not HBM)" % bps},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_decode(cfg, flat)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="decode", choices=["decode", "train"])
    ap.add_argument("--batch", type=int, default=20, help="utterances per GPU (reference decode_batch_size = 20, runQP.py:66)")
    ap.add_argument("--frames", type=int, default=2005, help="frames per utterance (2005 -> 10 s @22.05 kHz)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    rank, local, world = dist_setup(args.gpus)
    if args.mode == "decode":
        out = run_decode(args, rank, local, world)
    else:
        raise SystemExit("train mode: not built yet")
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
