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


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU, RCCL
    rendezvous on 127.0.0.1) BEFORE this process touches the GPU, relay their output and exit with their status."""
    if n_gpus <= 1 or "RANK" in os.environ:
        return
    import subprocess
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


BACKEND = "none"


def dist_setup(n_gpus):
    global BACKEND
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("QPN_BENCH_FORCE_PG"):
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(29500 + os.getpid() % 2000))):
            os.environ.setdefault(k, v)                # QPN_BENCH_FORCE_PG: one-rank process group (rehearses the RCCL calls on one GPU)
        if os.environ.get("QPN_BENCH_ONE_GPU"):      # dev aid: rehearse the N>1 code path with every rank on GPU 0
            local = 0
        torch.cuda.set_device(local)
        # every rank on one card cannot form an RCCL communicator (duplicate device): the one-GPU rehearsal uses gloo
        BACKEND = os.environ.get("QPN_DIST_BACKEND", "gloo" if (os.environ.get("QPN_BENCH_ONE_GPU") and world > 1) else "nccl")
        if BACKEND == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(BACKEND)
        assert dist.get_world_size() == world
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
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            d["_file"] = "profiles/" + name
            return d
        except Exception:
            continue
    return None


def measured_pmc():
    """matrix-core busy fraction per kernel (SQ_VALU_MFMA_BUSY_CYCLES pass, committed under profiles/): {rocprof kernel name: fraction}."""
    for name in ("r06_pmc_by_kernel.json", "r05_pmc_by_kernel.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            return {k: v.get("mfma_busy") for k, v in d.get("train", d).items() if isinstance(v, dict)}
        except Exception:
            continue
    return None


def decode_weight_bytes_per_sample(cfg):
    """172 B of per-sample inputs/outputs + 4 B x the parameters touched per sample
    (all but the one-hot table, of which two columns are looked up)."""
    C, Q = cfg.n_resch, cfg.n_quantize
    touched = cfg.n_params - Q * C * 2 + 2 * C - (cfg.upsampling_factor + 1)
    return 172 + 4 * touched


def cpu_baseline_decode(cfg, flat, n_frames):
    """The CPU oracle (a C port of the reference algorithm, validated against the reference's own streams) timed on this
    box's host cores as SURVEY §8d asks: B = 1 and B = 20 utterances of the bench length, greedy, threads over utterances
    (rows are independent; the port is scalar within an utterance) with all host cores and with 8 threads."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cpu_oracle
    from qpnet_amd import synth
    cpu_oracle.lib()
    ncores = os.cpu_count() or 1
    utts = [synth.decode_inputs(cfg, n_frames, 100 + b, 1.0) for b in range(20)]

    def one(u):
        x, h, d, n = u
        cpu_oracle.decode(cfg, flat, h, d, x, n)      # ctypes releases the GIL for the duration of the C call
        return n

    def timed(batch, threads):
        t0 = time.time()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            total = sum(ex.map(one, batch))
        return total / (time.time() - t0)

    b1 = timed(utts[:1], 1)
    b20_all = timed(utts, min(20, ncores))
    b20_8 = timed(utts, min(8, ncores))
    return {"value": b20_all, "unit": "samples/s", "cores": min(20, ncores), "kind": "port",
            "sample": "greedy decode of 20 x %d-frame utterances (%d samples each), one thread per utterance on %d host cores; same "
                      "synthetic features/weights as the GPU run" % (n_frames, utts[0][3], ncores),
            "batch1_one_core": b1, "batch20_8_threads": b20_8, "host_cores": ncores,
            "note": "reference code itself (torch CPU, 8 threads, survey container): 273 samples/s (BASELINE.md section 2)"}


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
    utts = [(1000 * rank + 100 + b, F, getattr(args, "f0_factor", 1.0)) for b in range(B)]
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
    tr = (measured_traffic() or {})
    hbm_meas = tr.get("decode", {}).get("hbm_bytes_per_sample")
    us = k_ms * 1e3 / max(ns)                      # device time per generated sample of one utterance (rows run concurrently)
    pipelined = B <= 48 and os.environ.get("QPN_DECODE_PIPE", "1") != "0"
    if pipelined:
        # five CUs per utterance, weights resident: per sample 4 hand-offs (0.421 us each inside an XCD) + 22 dependent stages
        # (LDS write -> barrier -> LDS read -> 16-FMA chain, 0.213 us each): profiles/r04_hop_microbench.txt (re-measured in round 4)
        floor, kernel, cus = 4 * 0.421 + 22 * 0.213, "k_decode_pipe", 5 * B
        bound = "latency (4 cross-CU hand-offs + 22 dependent matvec stages per sample)"
        note = ("five persistent workgroups (five CUs) per utterance -- fixed stack, adaptive stack, fixed-stack skip, skip + post-1, post-2 + pick -- "
                "every critical-path weight tile resident in VGPRs / LDS, hand-offs by 8-byte {tag, value} granules; floor = the "
                "microbenchmarked cost of the hand-offs and dependent stages of one sample (profiles/r04_hop_microbench.txt); "
                "HBM sees only the per-sample inputs/outputs (172 B algorithmic)")
    else:
        floor, kernel, cus = 11.65, "k_decode_fast" if cfg.n_resch <= 64 else "k_decode", B       # profiles/r01_l2_stream_floor.txt
        bound = "latency(L2 port)"
        note = ("one persistent workgroup (one CU) per utterance; weights (1.7 MB of tiles per sample) are re-streamed from L2 every "
                "sample (floor: one CU's 64 B/clk port), HBM sees only the per-sample inputs/outputs (172 B algorithmic)")
    out = {
        "metric": "AR decode samples/sec/GPU @22.05kHz (greedy)" if world == 1 else "AR decode samples/sec @22.05kHz (greedy), all GPUs",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config[3]: batch_fast_generate(argmax) of %d x %.1f s utterances per GPU, paper-size QPNet "
                               "(C=64,S=256,4F+4A), F=%d frames -> %d samples each" % (B, ns[0] / 22050.0, F, ns[0]),
                   "batch_per_gpu": B, "parallelism": "replicas x%d (no collective)" % world, "backend": BACKEND, "world_size": world},
        # the decode loop is a serial chain of dependent matvec stages per sample: its bound is latency, not HBM.
        # achieved / floor are microseconds per sample per utterance; frac = floor / achieved.
        "roofline": {"bound": bound, "achieved": us, "floor": floor, "peak": floor, "unit": "us/sample/utterance",
                     "frac": floor / us if us > 0 else 0.0,
                     "hbm_algorithmic": 172, "hbm_bytes_per_sample": hbm_meas,
                     # measured HBM bytes per generated sample / the 172 algorithmic ones: the write-through granules and polling loads of the hand-offs
                     "traffic_ratio": (hbm_meas / 172.0) if hbm_meas else None,
                     "hbm_achieved_GBps": 172.0 * sum(ns) / (k_ms * 1e-3) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS,
                     "traffic": hbm_meas * sum(ns) if hbm_meas else None,
                     "traffic_source": "%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, per sample x samples of this launch)" % tr.get("_file"),
                     "cus_busy": min(cus, 256) / 256.0,
                     "kernel": kernel, "kernel_ms": k_ms, "note": note},
    }
    out["plan"] = getattr(m, "last_decode_plan", "")
    if world == 1 and not args.no_cpu:
        # the reference decode script's default mode (softmax + draw, qpnet_decode.py:312-314): one untimed-in-`value` launch
        torch.cuda.synchronize(); t1 = time.perf_counter()
        m.batch_fast_generate(xb, hb, list(ns), bd, mode="sampling")
        torch.cuda.synchronize()
        out["sampling_mode"] = {"value": sum(ns) / (time.perf_counter() - t1), "unit": "samples/s", "kernel_ms": m.last_decode_kernel_ms}
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_decode(cfg, flat, F)
    return out


# profile groups of libqpnet_hip (include/qpnet_hip.h, QPN_PG_*): one per LAUNCH of the heavy kernels
PG_NAMES = ["prep+pack", "k_layer_fwd", "k_post_fwd", "k_ce", "k_post_bwd", "k_wgrad_w1", "k_layer_bwd", "grad_tail", "k_adam", "allreduce",
            "k_wgrad_wr", "k_wgrad_skip", "k_wgrad_post", "k_wgrad_causal"]
PG_WGRAD = (5, 10, 11, 12, 13)          # the five weight-gradient launches (the "k_wgrad" group of earlier rounds)


def pg_kernel_names(hoist):
    """rocprofv3's kernel names of the launches behind each profile group (paper-size geometry; profiles/r05_train_kernel_stats.csv)."""
    ks = 8 if hoist else 11
    return {1: "k_stack_fwd<%d>" % ks, 2: "k_post_fwd_w<5>", 4: "k_post_bwd_w<5>", 5: "k_wgrad3<3, 2, %d, false, 1>" % ks, 6: "k_stack_bwd<%d>" % ks,
            10: "k_wgrad3<0, 1, 4, false, 1>" if os.environ.get("QPN_STACK_QUEUE_BWD", "1") != "0" and os.environ.get("QPN_STACK_QUEUE", "1") != "0" else "k_wgrad3<0, 1, 4, true, 1>",     # (one A array behind the stack queue, which sums dXout's two parts in place)
            11: "k_wgrad3<0, 4, 4, false, 2>",      # (B = the gate product p.TH)
            12: "k_wgrad3<1, 4, 4, false, 2>", 13: "k_wgrad3<4, 1, 4, true, 1>"}


def train_flops(cfg, N1, BL, starts_out, hoist):
    """FLOPs (2 x MACs) of one training step per profile group, two ways:
      algorithmic -- SURVEY.md section 8d: per-row MACs of the reference's layers x exact per-layer row counts, padding excluded (the aux 1x1 at
                     SAMPLE rate, as the reference computes it);
      executed    -- what the matrix cores are asked to do here: with the auxiliary 1x1 hoisted to frame rate the gate contraction is
                     K = 2C plus one 4-deep step, its backward has no aux columns (+ eight 16x16x4 MFMAs per wave and tile for the frame-rate
                     accumulators), dW1 is 2C x 2C; without the hoist the aux columns are padded from 39 to 48.
    roofline fractions use the EXECUTED count (a kernel is not credited with work it no longer does)."""
    C, S, Q, A = cfg.n_resch, cfg.n_skipch, cfg.n_quantize, cfg.n_aux
    L = len(starts_out)
    rows = [N1 - s for s in starts_out]
    res = [C * C if i < L - 1 else 0 for i in range(L)]

    def groups(kf, kb, kw, extra_b):
        g = [0.0] * len(PG_NAMES)
        g[1] = 2.0 * sum(r * (kf * 2 * C + res[i]) for i, r in enumerate(rows))
        g[2] = 2.0 * BL * (L * C * S + S * S + S * Q)
        g[4] = 2.0 * BL * (Q * S + S * S + S * L * C)
        g[5] = 2.0 * sum(r * 2 * C * kw for r in rows)
        g[6] = 2.0 * sum(r * (res[i] + 2 * C * kb + extra_b) for i, r in enumerate(rows))
        g[10] = 2.0 * sum(r * res[i] for i, r in enumerate(rows))
        g[11] = 2.0 * BL * L * S * C
        g[12] = 2.0 * BL * (Q * S + S * S)
        return g
    algo = groups(2 * C + A, 2 * C + A, 2 * C + A, 0)
    if hoist:
        execd = groups(2 * C + 4, 2 * C, 2 * C, 32 * 16 * 16 * 4 // 16)      # (32 MFMAs of 16x16x4 per 16-row tile)
    else:
        ap = (A + 15) // 16 * 16
        execd = groups(2 * C + ap, 2 * C + ap, 2 * C + ap, 0)
    return algo, execd


def cpu_baseline_train(cfg, flat, batch):
    """torch-CPU port of the training step (oracle/train_torch.py: the time-major restatement of the reference's forward, torch autograd,
    torch.optim.Adam; pinned to the reference's fixture like the numpy oracle) timed on this box's host cores: a warm-up step, then
    full-size steps for ~15 s (bounded sample)."""
    import torch
    from oracle import train_torch as TT
    x, h, t, d, b = batch
    tr = TT.Trainer(cfg, flat)
    tr.step(x, h, t, d, b)                           # warm-up (thread pools, allocator)
    # the step's matrices are small (20 k x 64..256): more threads than ~16 make it slower on a many-core host.  One step at each
    # count, then the best one for the sample -- `cores` is the count actually used
    saved, sweep = torch.get_num_threads(), {}
    for nt in sorted({c for c in (8, 16, 32, saved) if c <= saved}):
        torch.set_num_threads(nt)
        t0 = time.time(); tr.step(x, h, t, d, b); sweep[nt] = time.time() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 < 10.0 and n < 50):
        tr.step(x, h, t, d, b); n += 1
    dt = (time.time() - t0) / n
    torch.set_num_threads(saved)
    return {"value": 1.0 / dt, "unit": "steps/s", "cores": best, "kind": "port",
            "threads_tried_s_per_step": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": "%d full-size steps (forward+CE+backward+Adam, chunk of %d samples) of the torch-CPU float32 port, after one warm-up step" % (n, x.shape[1]),
            "note": "the reference's own torch-CPU step: 1.20 s/step on 8 threads in the survey container (BASELINE.md section 2)"}


def dropin_loop_rate(m, cfg, batches, nchunks, flat_adam, steps=200):
    """The reference trainer's loop, unchanged, on the drop-in module (src/bin/qpnet_train.py:517-531):
    model(x,h,d,b) -> nn.CrossEntropyLoss -> zero_grad -> backward -> Adam.step."""
    import torch
    from qpnet_amd.train import FlatAdam
    opt = FlatAdam(m, lr=1e-4) if flat_adam else torch.optim.Adam(m.parameters(), lr=1e-4)
    crit = torch.nn.CrossEntropyLoss()

    def step(i):
        x, h, t, d, b = batches[i % nchunks]
        out = m(x, h, d, b)
        BL = out.shape[1]
        loss = crit(out.view(-1, cfg.n_quantize), t[:, -BL:].reshape(-1))
        opt.zero_grad()
        loss.backward()
        opt.step()
    # (torch loads the code objects of its own loss kernels lazily: the first cross_entropy costs ~95 ms, the eighth backward ~80 ms on this image -- both
    #  inside a 30-step window made the first leg read 230-310 steps/s)
    for i in range(12):
        step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


def runner_loop_rate(m, tr, cfg, dev, steps=300, sync_loss=False):
    """What `python -m qpnet_amd.run_train` does per iteration (runners.run_train: loaders.train_generator -> PinnedStager -> a depth-2
    prefetch thread -> FusedTrainer.step(want_loss="lagged"): EVERY step's loss reaches the host, as the reference's loss.item() does
    (src/bin/qpnet_train.py:533), one step late and without draining the stream; sync_loss: read in-step, the reference's literal order) on
    an in-memory corpus of synthetic utterances of VCC2018 shape."""
    import torch
    from qpnet_amd import loaders, synth
    from qpnet_amd.runners import PinnedStager, Prefetcher
    U = cfg.upsampling_factor
    rs = np.random.RandomState(0)
    utts = []
    for i in range(24):
        nf = int(rs.randint(600, 1200))                 # 3-6 s utterances
        utts.append((rs.uniform(-1, 1, nf * U + 5).astype(np.float32), synth.make_features(nf, 400 + i, 45.0, 300.0)))
    mean, scale = synth.scaler_stats()
    np.random.seed(1)
    gen = loaders.train_generator(utts, cfg.receptiveCausal_field, cfg.receptiveF_field, cfg.receptiveA_field, 22050,
                                  wav_transform=loaders.mu_law_transform(cfg.n_quantize), feat_transform=lambda h: (h - mean) / scale,
                                  batch_length=20000, max_length=30000, upsampling_factor=U, shuffle=True)
    stage = PinnedStager(dev)

    def batches():
        for bx, bh, bt, bd, bb in gen:
            dv = stage({"x": bx, "h": bh, "t": bt, "d": bd})
            yield dv["x"], dv["h"], dv["t"], dv["d"], bb, int(np.ceil(float(bd.max())))
    stream = Prefetcher(batches())
    mode = True if sync_loss else "lagged"
    for _ in range(5):
        bx, bh, bt, bd, bb, maxd = next(stream)
        tr.step(bx, bh, bt, bd, bb, want_loss=mode, maxd=maxd)
    tr.flush_loss()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    total, got = 0.0, 0
    w_batch = w_step = 0.0
    for i in range(steps):
        ta = time.perf_counter()
        bx, bh, bt, bd, bb, maxd = next(stream)
        tb = time.perf_counter()
        v = tr.step(bx, bh, bt, bd, bb, want_loss=mode, maxd=maxd)
        w_batch += tb - ta; w_step += time.perf_counter() - tb
        if v is not None:
            total += v; got += 1
        if not sync_loss and (i + 1) % 100 == 0:          # (run_train's default reporting interval)
            v = tr.flush_loss()
            if v is not None:
                total += v; got += 1
    v = None if sync_loss else tr.flush_loss()
    if v is not None:
        total += v; got += 1
    torch.cuda.synchronize()
    assert got == steps and np.isfinite(total)
    dt = time.perf_counter() - t0
    # where the main thread's iteration went (us): waiting for the loader thread's batch / inside step() -- on a box whose host is busy the loop falls from ~1440 to
    # ~1185 steps/s (five runs on five boxes in round 6; tools/runner_loop_split.py has the loader thread's side)
    runner_loop_rate.split = {"wait_batch_us": w_batch / steps * 1e6, "in_step_us": w_step / steps * 1e6, "iteration_us": dt / steps * 1e6}
    return steps / dt


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
    tr = FusedTrainer(m, lr=1e-4, world_size=2 if (world == 1 and os.environ.get("QPN_EXCHANGE_ALWAYS")) else world)
    if world > 1:
        from qpnet_amd import parallel
        from qpnet_amd.train import ensure_flat
        parallel.broadcast_parameters(ensure_flat(m, dev))          # every rank starts from rank 0's parameters
    # utterance-sharded synthetic chunks: rank r consumes chunks r, r+world, ...
    nchunks = 4
    # SURVEY 8d's config[2] shape: batch_length 20000 / max_length 30000, the corpus' pitch floor 45 Hz in every chunk -> ceil(max d) 62,
    # RF 946, (RF + BL) % 110 trimmed: 20 900 samples per chunk, 19 954 output rows
    host_batches = [synth.train_inputs(cfg, 20000, 5000 + 17 * (rank + world * i), 30000, f0_lo=45.0, f0_hi=300.0, pin_f0_floor=True) for i in range(nchunks)]
    batches = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in hb] for hb in host_batches]
    # the loader knows ceil(max d) of a chunk on the host (it built d there), so the step needs no device read-back
    maxds = [int(np.ceil(hb[3]).max()) for hb in host_batches]
    blens = [hb[4] for hb in host_batches]

    def step(i, want_loss=False):
        x, h, t, d, _ = batches[i % nchunks]
        return tr.step(x, h, t, d, blens[i % nchunks], want_loss=want_loss, maxd=maxds[i % nchunks])

    # Initialisation, outside the W warm-up steps and the K timed ones: the first step of a handle loads the code objects and allocates the arena (12-38 ms
    # of host time with the device idle), and a device that has idled for that long runs its next ~25 steps 5-8 % slow, whoever launched them (measured:
    # tools/warmup_curve.py, MEASUREMENTS R6.1 -- every matrix kernel alike, no effect of lr = 0, of a pre-warmed second handle or of which kernels ran
    # before the pause: the power state after the idle period, not the library's state).  The trainer is therefore stepped until the device has been
    # busy for QPN_BENCH_INIT_MS (default 40 ms) before the contract's warm-up starts: `--steps 20 --warmup 5` then measures the steady rate a run is in
    # from its first 30 ms on, not the ramp.  config.init reports what was run.
    init_ms = float(os.environ.get("QPN_BENCH_INIT_MS", "40"))
    n_init = 0
    if init_ms > 0:
        step(0); torch.cuda.synchronize(); n_init = 1
        t_init = time.perf_counter()
        while (time.perf_counter() - t_init) * 1e3 < init_ms:
            step(n_init); n_init += 1
    for i in range(args.warmup):
        step(i)
    barrier(world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    # per-kernel-group device time of more (untimed) steps: serial (every launch alone on one stream) and as the timed loop runs them (two streams)
    L_, hd = m._native(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    nprof = 3

    def profile(begin):
        buf = (C.c_float * len(PG_NAMES))()
        _lib.check(begin(hd, stream))
        for i in range(nprof):
            step(i)
        _lib.check(L_.qpn_train_profile_end(hd, buf, len(PG_NAMES), stream))
        return [v / nprof for v in buf]
    ms = profile(L_.qpn_train_profile_begin)
    ms_ov = profile(L_.qpn_train_profile_begin_overlapped) if world == 1 else ms
    x0, h0, t0_, d0, b0 = host_batches[0]
    BL = int(b0[0]); maxd = int(np.ceil(d0).max())
    N1 = cfg.receptive_field(maxd) + BL - 1
    starts, s_ = [], 0
    for dil in cfg.dilationsF:
        s_ += dil; starts.append(s_)
    for dil in cfg.dilationsA:
        s_ += dil * maxd; starts.append(s_)
    hoist = os.environ.get("QPN_AUX_HOIST", "1") != "0"
    fl_algo, fl = train_flops(cfg, N1, BL, starts, hoist)
    knames = pg_kernel_names(hoist)
    if os.environ.get("QPN_POST_FUSE", "1") != "0":
        # the fused step runs the post-net's forward (with the cross entropy) and backward of a row tile as ONE kernel: its time is in the forward's group
        knames[2] = "k_post_fb_w<5>"; del knames[4]
        fl = list(fl); fl_algo = list(fl_algo)
        fl[2] += fl[4]; fl[4] = 0.0; fl_algo[2] += fl_algo[4]; fl_algo[4] = 0.0
    tr_meas = measured_traffic() or {}
    pmc = measured_pmc() or {}
    def kernel_list(msv):
        out_ = []
        for gidx, kname in knames.items():
            if msv[gidx] <= 0:
                continue
            tfl = fl[gidx] / (msv[gidx] * 1e-3) / 1e12
            out_.append({"name": kname, "group": PG_NAMES[gidx], "us": round(msv[gidx] * 1e3, 1), "gflop": round(fl[gidx] / 1e9, 3),
                         "gflop_algorithmic": round(fl_algo[gidx] / 1e9, 3), "tflops": round(tfl, 1), "frac": round(tfl / F32_MFMA_PEAK_TFLOPS, 3),
                         "mfma_busy": pmc.get(kname), "hbm_bytes": tr_meas.get("train", {}).get("hbm_bytes_by_kernel", {}).get(kname)})
        out_.sort(key=lambda k: -k["us"])
        return out_
    kernels = kernel_list(ms)                      # every launch alone (one stream)
    overlapped = kernel_list(ms_ov)                # ... and inside the step as the timed loop runs it (two streams): mfma_busy / hbm_bytes are the serial PMC passes' (per launch: unchanged by the overlap)
    dom = overlapped[0]                            # the single longest kernel of the step AS IT RUNS (per-launch HIP events on the stream of the launch)
    low = min((k_ for k_ in kernels if k_["gflop"] > 0 and k_["us"] >= 50.0), key=lambda k_: k_["frac"], default=dom)      # ... and, of the kernels of 50 us and more, the one furthest below the roofline when alone
    low_ov = min((k_ for k_ in overlapped if k_["gflop"] > 0 and k_["us"] >= 50.0), key=lambda k_: k_["frac"], default=dom)
    wg_ms = sum(ms[i] for i in PG_WGRAD); wg_fl = sum(fl[i] for i in PG_WGRAD)
    total_flops, total_algo = sum(fl), sum(fl_algo)
    value = args.steps * world / dt
    out = {
        "metric": "train steps/sec (batch-1 chunk-steps of RF+20000 samples, aggregate over GPUs)",
        "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config[1]: paper-size SI-QPNet (C=64,S=256,4F+4A) training step, forward+CE+backward+Adam on one chunk "
                               "of %d samples (RF %d + batch_length %d), batch 1 per GPU" % (x0.shape[1], N1 + 1 - BL, BL),
                   "global_batch": world, "parallelism": "dp%d (utterance-sharded chunks, one flat-gradient all-reduce per step)" % world,
                   "backend": BACKEND, "world_size": (dist.get_world_size() if dist.is_initialized() else 1),
                   "aux_1x1": "frame rate (hoisted: K = 128 + one 4-deep step)" if hoist else "sample rate (K = 176)",
                   "init": {"steps": n_init, "busy_ms": init_ms, "why": "handle initialisation + the device's ramp out of its idle power state, before the W warm-up steps (MEASUREMENTS R6.1)"}},
        "roofline": {"bound": "mfma", "achieved": dom["tflops"], "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": dom["tflops"] / F32_MFMA_PEAK_TFLOPS,
                     "traffic": dom["hbm_bytes"],
                     "traffic_source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, per kernel and launch; commit %s)"
                                       % (tr_meas.get("_file"), tr_meas.get("commit", "?")),
                     "step_traffic": tr_meas.get("train", {}).get("hbm_bytes_per_step_all_kernels"),
                     "kernel": dom["name"], "kernel_ms": dom["us"] / 1e3, "flops_per_launch": dom["gflop"] * 1e9,
                     # every heavy kernel of the step, longest first: us from HIP events around the launch (one stream), gflop = EXECUTED FLOPs,
                     # mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES fraction from the committed PMC pass (profiles/), hbm_bytes per launch likewise
                     "kernels": kernels[:8],
                     # the same launches timed INSIDE the two-stream step (qpn_train_profile_begin_overlapped): roofline.kernel / achieved / frac are the longest of THESE;
                     # the committed rocprofv3 pass of the same two-stream command is profiles/r06_train2_kernel_stats.csv
                     "overlapped": overlapped[:8],
                     "kernel_alone": {"name": kernels[0]["name"], "us": kernels[0]["us"], "frac": kernels[0]["frac"]},
                     # the kernel (of those >= 50 us) furthest below the matrix-core roofline, alone and inside the step
                     "lowest": {"name": low["name"], "us": low["us"], "frac": low["frac"], "mfma_busy": low["mfma_busy"]},
                     "lowest_overlapped": {"name": low_ov["name"], "us": low_ov["us"], "frac": low_ov["frac"]},
                     "wgrad_group": {"launches": 5, "ms": round(wg_ms, 4), "tflops": round(wg_fl / (wg_ms * 1e-3) / 1e12, 1) if wg_ms > 0 else None,
                                     "frac": round(wg_fl / (wg_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 3) if wg_ms > 0 else None},
                     "step_tflops": total_flops / (sum(ms) * 1e-3) / 1e12, "step_device_ms": sum(ms),
                     # the whole step against the same peak: executed FLOPs of every group / the timed loop's ms_per_step
                     "step_flops": total_flops, "step_flops_algorithmic": total_algo, "step_achieved": total_flops / (dt / args.steps) / 1e12,
                     "step_frac": total_flops / (dt / args.steps) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                     "step_frac_algorithmic": total_algo / (dt / args.steps) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                     "groups_ms": dict(zip(PG_NAMES, [round(v, 4) for v in ms])),
                     "groups_ms_overlapped": dict(zip(PG_NAMES, [round(v, 4) for v in ms_ov])),
                     "note": "roofline.kernel = the single longest kernel of the step AS THE TIMED LOOP RUNS IT (two streams: the skip / post-net weight gradients, the early slab "
                             "reduction, the aux-gradient tail and dWr on a side stream next to the stack backward and dW1); achieved = its EXECUTED FLOPs / its device time "
                             "from HIP events around the launch on the stream it runs on (roofline.overlapped); roofline.kernels = the same launches alone on one stream.  "
                             "step_frac counts executed FLOPs over the timed loop's ms_per_step; step_frac_algorithmic credits the reference's sample-rate aux 1x1 (SURVEY 8d) instead."},
    }
    if world == 1 and not args.no_cpu:
        # the drop-in path north_star describes: the reference's own loop on this module (weights keep training; timing only)
        # the loop `python -m qpnet_amd.run_train` runs: generator + pinned staging + prefetch thread + step(want_loss="lagged"): every step's loss
        # reaches the host one step late; ..._sync_loss: read inside the step, the reference's literal order (QPN_RUN_TRAIN_SYNC_LOSS=1)
        out["runner_loop_steps_per_s"] = runner_loop_rate(m, tr, cfg, dev)
        out["runner_loop_split"] = dict(runner_loop_rate.split)
        try:
            out["runner_loop_split"]["host_loadavg"] = open("/proc/loadavg").read().split()[:3]
        except OSError:
            pass
        out["runner_loop_sync_loss_steps_per_s"] = runner_loop_rate(m, tr, cfg, dev, steps=150, sync_loss=True)
        out["dropin_loop_steps_per_s"] = dropin_loop_rate(m, cfg, batches, nchunks, flat_adam=False)
        out["dropin_loop_flat_adam_steps_per_s"] = dropin_loop_rate(m, cfg, batches, nchunks, flat_adam=True)
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_train(cfg, flat, host_batches[0])
    return out


def run_default_geometry(local):
    """SURVEY 8d "also report D": the geometry runQP.py instantiates (n_resch 512, 12 fixed + 4 adaptive layers, 24.2 M
    parameters; src/utils/param_model.py:58-64) -- training step on the LDS-tiled GEMM path (train_gemm.hip) and decode on
    the cooperative multi-workgroup kernels (decode_coop.hip; batches above 16 rows: decode_coopb.hip).  A few steps / 500-frame utterances: extra keys, not `value`."""
    import ctypes as C
    import torch
    from qpnet_amd import synth, _lib
    from qpnet_amd.config import DEFAULT
    from qpnet_amd.qpnet import QPNet
    from qpnet_amd.train import FusedTrainer
    dev = torch.device("cuda", local)
    cfg = DEFAULT
    flat = synth.make_weights(cfg, 13)
    m = QPNet(**cfg.kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
    m = m.to(dev).train()
    tr = FusedTrainer(m, lr=1e-4)
    hb = [synth.train_inputs(cfg, 20000, 5000 + 17 * i, 30000, f0_lo=55.0, f0_hi=300.0) for i in range(2)]
    bt = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in b] for b in hb]
    maxds = [int(np.ceil(b[3]).max()) for b in hb]

    def step(i):
        x, h, t, d, _ = bt[i % 2]
        return tr.step(x, h, t, d, hb[i % 2][4], want_loss=False, maxd=maxds[i % 2])
    for i in range(2):
        step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 8
    for i in range(n):
        step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    L_, hd = m._native(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ms = (C.c_float * len(PG_NAMES))()
    _lib.check(L_.qpn_train_profile_begin(hd, stream))
    for i in range(2):
        step(i)
    _lib.check(L_.qpn_train_profile_end(hd, ms, len(PG_NAMES), stream))
    ms = [v / 2 for v in ms]
    BL = int(hb[0][4][0]); maxd = maxds[0]; N1 = cfg.receptive_field(maxd) + BL - 1
    starts, s_ = [], 0
    for dil in cfg.dilationsF:
        s_ += dil; starts.append(s_)
    for dil in cfg.dilationsA:
        s_ += dil * maxd; starts.append(s_)
    fl, _ = train_flops(cfg, N1, BL, starts, False)          # algorithmic FLOPs; the GEMM path marks all its weight gradients as one group
    fl[5] = sum(fl[i] for i in PG_WGRAD)
    for i in PG_WGRAD[1:]:
        fl[i] = 0.0
    out = {"geometry": "repo default: C=512, S=256, F=[1,2,4,8]x3, A=[1,2,4,8], 24151151 parameters",
           "train": {"steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "tflop_per_step": sum(fl) / 1e12,
                     "roofline": {"bound": "mfma", "achieved": sum(fl) / dt / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": sum(fl) / dt / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                  "groups_ms": dict(zip(PG_NAMES, [round(v, 3) for v in ms])),
                                  "groups_tflops": {k: round(f / (v * 1e-3) / 1e12, 1) for k, v, f in zip(PG_NAMES, ms, fl) if f > 0 and v > 0}},
                     "reference_cpu_s_per_step": 17.9}}
    del tr, bt
    m = m.eval()
    dec = {}
    FR = 500                                   # 2.5 s utterances: 54 999 samples each
    Lz = len(cfg.dilationsF) + len(cfg.dilationsA)
    n_edges = 2 * Lz + 3                       # dependent all-gathers per generated sample: gate vector + block output per layer, 3 in the post-net
    w_bytes = 4.0 * (cfg.n_params - cfg.n_quantize * cfg.n_resch * 2 + 2 * cfg.n_resch)      # weights touched per sample (two rows of the one-hot table)
    stream_us = w_bytes / 8.6e12 * 1e6         # gathered reads from the Infinity Cache, whole chip (MI355X_MICROARCH.md: 8.6 TB/s)
    hop_us = 1.46                              # one all-gather edge of this kernel's transport, measured (tools/allgather_floor.hip, profiles/r04_allgather_floor.txt)
    for B in (1, 20):
        bx, bh, bd, ns = synth.decode_batch(cfg, [(100 + b, FR, 1.0) for b in range(B)])
        xb, hbt = torch.from_numpy(bx).to(dev), torch.from_numpy(bh).to(dev)
        if B == 1:
            wx, wh, wd, wn = synth.decode_batch(cfg, [(100, 10, 1.0)])
            m.batch_fast_generate(torch.from_numpy(wx).to(dev), torch.from_numpy(wh).to(dev), list(wn), wd, mode="argmax")     # warm-up (allocations)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.batch_fast_generate(xb, hbt, list(ns), bd, mode="argmax")
        torch.cuda.synchronize(); dtd = time.perf_counter() - t0
        us = m.last_decode_kernel_ms * 1e3 / max(ns)
        floor = n_edges * hop_us + stream_us          # the hops are serial; at B = 1 the stream of one step has nothing else to hide behind
        plan = getattr(m, "last_decode_plan", "")
        batched = plan.startswith("coopb ")
        streams = int(plan.split("groups=")[1].split()[0]) if batched else B      # weight streams per sample step: one per group of workgroups
        if batched:
            note = ("k_decode_coopb (decode_coopb.hip): the utterances are the N dimension of v_mfma_f32_16x16x4_f32 (four chained MFMAs = the spec's 16-term fma chain, "
                    "bit for bit; chunk accumulators added in the spec's tree order), a group of n_resch / 8 workgroups serves up to 16 utterances and streams the "
                    "weights once per sample step for all of them; floor = (2L+3) x 1.46 us (profiles/r04_allgather_floor.txt) + 96.6 MB / 8.6 TB/s. "
                    "What binds: the 2L+3 publish -> visible -> gathered -> barrier edges of a step (phase times: profiles/r06_coopb_phases.txt), whose gathers "
                    "grow with the utterances of the group; per-utterance kernel at this batch: profiles/r06_coopb_batches.txt")
        else:
            note = ("floor = (2L+3) x 1.46 us (the kernel's own all-gather edge with no arithmetic and no weight stream, measured: "
                    "profiles/r04_allgather_floor.txt -- 51 us per sample, i.e. above the 45.35 us of real time before a weight is read) + "
                    "96.6 MB / 8.6 TB/s (Infinity-Cache gather rate, whole chip)")
        dec["batch%d" % B] = {"samples_per_s": sum(ns) / dtd, "us_per_sample_per_utterance": us, "real_time_factor": (sum(ns) / dtd / B) / 22050.0,
                              "plan": plan, "kernel": "k_decode_coopb" if batched else "k_decode_coop",
                              "workload": "%d x %d-frame utterances (%d samples each)" % (B, FR, ns[0]),
                              "roofline": {"bound": "latency: %d dependent all-gathers per sample + the weight stream" % n_edges,
                                           "achieved": us, "floor": floor, "peak": floor, "unit": "us/sample/utterance", "frac": floor / us,
                                           "weight_stream_floor_us": stream_us, "handoff_floor_us": n_edges * hop_us,
                                           "weights_MB_per_sample": w_bytes / 1e6, "weight_streams_per_step": streams,
                                           "weight_stream_achieved_TBps": w_bytes * streams / (us * 1e-6) / 1e12,
                                           "note": note}}
    dec["reference_cpu_samples_per_s"] = 40
    dec["kernel"] = ("up to 16 rows: k_decode_coop (G workgroups per utterance, G = largest power of two with B*G <= CUs whose row slices are whole tiles); "
                     "above: k_decode_coopb (the rows batched into the MFMA's N dimension, n_resch / 8 workgroups per group of <= 16 rows)")
    out["decode"] = dec
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
    ap.add_argument("--f0-factor", dest="f0_factor", type=float, default=1.0, help="F0 scaling of the decode workload (BASELINE config[4]: 0.5 / 1.5)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 3 if args.mode == "decode" else 200
    if args.warmup is None:
        args.warmup = 1 if args.mode == "decode" else 20
    self_launch(args.gpus)
    rank, local, world = dist_setup(args.gpus)
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: measuring %d rank(s)" % (args.gpus, world, world), file=sys.stderr)
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
                                       "kernel_ms": d1["roofline"]["kernel_ms"], "kernel": d1["roofline"]["kernel"]}
            # BASELINE config[4]: F0-scaled decode (0.5x / 1.5x pitch: the per-sample dynamic dilation gather at its deepest / shallowest),
            # one utterance per GPU
            out["decode"]["f0_scaled"] = {}
            for fac in (0.5, 1.5):
                a5 = copy.copy(a3); a5.f0_factor = fac; a5.steps, a5.warmup = 1, 0
                d5 = run_decode(a5, rank, local, world)
                out["decode"]["f0_scaled"]["x%.1f" % fac] = {"value": d5["value"], "unit": d5["unit"], "us_per_sample_per_utterance": d5["roofline"]["achieved"],
                                                              "kernel": d5["roofline"]["kernel"]}
            # decode_batch_size is the caller's choice (--batch_size, src/bin/qpnet_decode.py:52): the chip full (48 five-role groups x 5 CUs)
            # and beyond (a group steps two utterances alternately; plan reported)
            out["decode"]["larger_batches"] = {}
            for Bx in (48, 49, 64, 96):
                a6 = copy.copy(a3); a6.batch = Bx; a6.steps, a6.warmup = 1, 0
                d6 = run_decode(a6, rank, local, world)
                out["decode"]["larger_batches"]["batch%d" % Bx] = {"value": d6["value"], "unit": d6["unit"], "ms_per_step": d6["ms_per_step"],
                                                                  "plan": d6.get("plan")}
            os.environ["QPN_DECODE_PIPE"] = "0"          # the one-CU-per-utterance kernel on the same workload, for comparison
            try:
                a4 = copy.copy(a2); a4.no_cpu = True; a4.steps, a4.warmup = 1, 0
                d20 = run_decode(a4, rank, local, world)
                out["decode"]["one_cu_per_utterance"] = {"value": d20["value"], "unit": d20["unit"], "us_per_sample_per_utterance": d20["roofline"]["achieved"],
                                                         "kernel": d20["roofline"]["kernel"], "frac_of_l2_stream_floor": d20["roofline"]["frac"]}
            finally:
                os.environ.pop("QPN_DECODE_PIPE", None)
            if not args.no_cpu:
                try:
                    out["default_geometry"] = run_default_geometry(local)
                except Exception as e:        # an extra: never let it take the headline line down
                    out["default_geometry"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
