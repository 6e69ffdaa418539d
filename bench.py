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
        if os.environ.get("QPN_BENCH_ONE_GPU"):      # dev aid: rehearse the N>1 code path with every rank on GPU 0 (gloo collectives)
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
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


def measured_traffic():
    """HBM traffic measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (committed under profiles/)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            return json.load(f)
    except Exception:
        return None


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
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (measured_traffic() or {}).get("decode", {}).get("hbm_bytes_per_sample", 0) * sum(ns) or None,
                     "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, per sample x samples of this launch)",
                     "kernel": "k_decode_fast" if cfg.n_resch <= 64 else "k_decode", "kernel_ms": k_ms,
                     "l2_stream_floor_us_per_sample": 11.65,
                     "note": "algorithmic bytes = (172 + 4*params touched) B/sample = %d B/sample (SURVEY 8d, weights re-streamed "
                             "every sample; they are L2-resident so this is L2->CU traffic, not HBM)" % bps},
    }
    if world == 1 and not args.no_cpu:
        # the reference decode script's default mode (softmax + draw, qpnet_decode.py:312-314): one untimed-in-`value` launch
        torch.cuda.synchronize(); t1 = time.perf_counter()
        m.batch_fast_generate(xb, hb, list(ns), bd, mode="sampling")
        torch.cuda.synchronize()
        out["sampling_mode"] = {"value": sum(ns) / (time.perf_counter() - t1), "unit": "samples/s", "kernel_ms": m.last_decode_kernel_ms}
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_decode(cfg, flat)
    return out


PG_NAMES = ["prep+pack", "k_layer_fwd", "k_post_fwd", "k_ce", "k_post_bwd", "k_wgrad", "k_layer_bwd", "grad_tail", "k_adam"]


def train_flops(cfg, N1, BL, starts_out):
    """Algorithmic FLOPs (2 x MACs, padding excluded) of one training step per kernel group
    (per-row MACs x exact per-layer row counts)."""
    C, S, Q, A = cfg.n_resch, cfg.n_skipch, cfg.n_quantize, cfg.n_aux
    L = len(starts_out)
    Kt = 2 * C + A
    rows = [N1 - s for s in starts_out]
    lf = sum(r * (Kt * 2 * C + (C * C if i < L - 1 else 0)) for i, r in enumerate(rows))
    pf = BL * (L * C * S + S * S + S * Q)
    pb = BL * (Q * S + S * S + S * L * C)
    wg = BL * (Q * S + S * S + L * S * C) + sum(r * (2 * C * Kt + (C * C if i < L - 1 else 0)) for i, r in enumerate(rows))
    lb = sum(r * ((C * C if i < L - 1 else 0) + 2 * C * Kt) for i, r in enumerate(rows))
    g = [0.0] * len(PG_NAMES)
    g[1], g[2], g[4], g[5], g[6] = 2.0 * lf, 2.0 * pf, 2.0 * pb, 2.0 * wg, 2.0 * lb
    return g


def cpu_baseline_train(cfg, flat, batch):
    """numpy float32 port of the training step (oracle/train_oracle.py, pinned to the reference's
    autograd) timed on this box's host cores: ONE full-size step (bounded sample)."""
    from oracle import train_oracle as TO
    x, h, t, d, b = batch
    w = flat.copy()
    opt = TO.Adam(w.size)
    t0 = time.time()
    TO.train_step(cfg, w, opt, x, h, t, d, b)
    dt = time.time() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count()
    return {"value": 1.0 / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "one full-size step (forward+CE+backward+Adam, chunk of %d samples) of the numpy float32 oracle" % x.shape[1]}


def run_train(args, rank, local, world):
    import ctypes as C
    import torch
    import torch.distributed as dist
    from qpnet_amd import synth, harness, _lib
    from qpnet_amd.config import PAPER
    from qpnet_amd.qpnet import QPNet
    from qpnet_amd.train import FusedTrainer
    dev = torch.device("cuda", local)
    cfg = PAPER
    flat = synth.make_weights(cfg, 13)
    m = QPNet(**cfg.kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
    m = m.to(dev).train()
    tr = FusedTrainer(m, lr=1e-4, world_size=world)
    # utterance-sharded synthetic chunks: rank r consumes chunks r, r+world, ...
    nchunks = 4
    host_batches = [synth.train_inputs(cfg, 20000, 5000 + 17 * (rank + world * i), 30000, f0_lo=55.0, f0_hi=300.0) for i in range(nchunks)]
    batches = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in hb] for hb in host_batches]
    # the loader knows ceil(max d) of a chunk on the host (it built d there), so the step needs no device read-back
    maxds = [int(np.ceil(hb[3]).max()) for hb in host_batches]
    blens = [hb[4] for hb in host_batches]

    def step(i, want_loss=False):
        x, h, t, d, _ = batches[i % nchunks]
        return tr.step(x, h, t, d, blens[i % nchunks], want_loss=want_loss, maxd=maxds[i % nchunks])

    for i in range(args.warmup):
        step(i)
    barrier(world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    # per-kernel-group device time of one more (untimed) step -> roofline of the dominant group
    L_, hd = m._native(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ms = (C.c_float * len(PG_NAMES))()
    nprof = 3
    _lib.check(L_.qpn_train_profile_begin(hd, stream))
    for i in range(nprof):
        step(i)
    _lib.check(L_.qpn_train_profile_end(hd, ms, len(PG_NAMES), stream))
    ms = [v / nprof for v in ms]
    x0, h0, t0_, d0, b0 = host_batches[0]
    BL = int(b0[0]); maxd = int(np.ceil(d0).max())
    N1 = cfg.receptive_field(maxd) + BL - 1
    starts, s_ = [], 0
    for dil in cfg.dilationsF:
        s_ += dil; starts.append(s_)
    for dil in cfg.dilationsA:
        s_ += dil * maxd; starts.append(s_)
    fl = train_flops(cfg, N1, BL, starts)
    dom = int(np.argmax(ms))
    achieved = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    total_flops = sum(fl)
    value = args.steps * world / dt
    out = {
        "metric": "train steps/sec (batch-1 chunk-steps of RF+20000 samples, aggregate over GPUs)",
        "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config[1]: paper-size SI-QPNet (C=64,S=256,4F+4A) training step, forward+CE+backward+Adam on one chunk "
                               "of %d samples (RF %d + batch_length %d), batch 1 per GPU" % (x0.shape[1], N1 + 1 - BL, BL),
                   "global_batch": world, "parallelism": "dp%d (utterance-sharded chunks, one flat-gradient all-reduce per step)" % world},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / F32_MFMA_PEAK_TFLOPS,
                     "traffic": (measured_traffic() or {}).get("train", {}).get("hbm_bytes_per_step") if PG_NAMES[dom] == "k_wgrad" else None,
                     "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, the weight-gradient launches of one step)",
                     "kernel": PG_NAMES[dom],
                     "kernel_ms": ms[dom], "flops_per_launch_group": fl[dom],
                     "step_tflops": total_flops / (sum(ms) * 1e-3) / 1e12, "step_device_ms": sum(ms),
                     "groups_ms": dict(zip(PG_NAMES, [round(v, 4) for v in ms])),
                     "note": "achieved = algorithmic FLOPs of the dominant kernel group (all its launches in one step) / its summed "
                             "device time from HIP events on the launch stream; step_tflops = whole step"},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_train(cfg, flat, host_batches[0])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", default="all", choices=["all", "decode", "train"],
                    help="all (default) = train step as the headline value (BASELINE config[1]) + a 'decode' object (config[3]) at N=1")
    ap.add_argument("--batch", type=int, default=20, help="utterances per GPU (reference decode_batch_size = 20, runQP.py:66)")
    ap.add_argument("--frames", type=int, default=2005, help="frames per utterance (2005 -> 10 s @22.05 kHz)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 3 if args.mode == "decode" else 50
    if args.warmup is None:
        args.warmup = 1 if args.mode == "decode" else 5
    rank, local, world = dist_setup(args.gpus)
    if args.mode == "decode":
        out = run_decode(args, rank, local, world)
    else:
        out = run_train(args, rank, local, world)
        if args.mode == "all" and world == 1:
            # the other half of BASELINE.json's metric: AR decode samples/sec/GPU (same JSON line)
            import copy
            a2 = copy.copy(args); a2.steps, a2.warmup = 2, 1
            out["decode"] = run_decode(a2, rank, local, world)
            a3 = copy.copy(a2); a3.batch, a3.no_cpu = 1, True
            d1 = run_decode(a3, rank, local, world)
            out["decode"]["batch1"] = {"value": d1["value"], "unit": d1["unit"], "ms_per_step": d1["ms_per_step"],
                                       "kernel_ms": d1["roofline"]["kernel_ms"]}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
