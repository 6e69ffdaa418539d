"""Dev: where an iteration of the run_train loop (bench.py runner_loop_rate) spends its host time -- waiting for the loader thread's batch vs inside FusedTrainer.step --
and what the box is (CPU model, clock, cores).   python tools/runner_loop_split.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from qpnet_amd import loaders, synth
    from qpnet_amd.config import PAPER as cfg
    from qpnet_amd.qpnet import QPNet
    from qpnet_amd.runners import PinnedStager, Prefetcher
    from qpnet_amd.train import FusedTrainer
    try:
        info = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        mhz = [float(l.split(":")[1]) for l in open("/proc/cpuinfo") if l.startswith("cpu MHz")]
        print("host: %s x %d, %.0f-%.0f MHz now, affinity %d cpus, load %s" % (info[0], len(info), min(mhz), max(mhz), len(os.sched_getaffinity(0)), open("/proc/loadavg").read().strip()))
    except Exception as e:
        print("host: ?", e)
    dev = torch.device("cuda", 0)
    flat = synth.make_weights(cfg, 13)
    m = QPNet(**cfg.kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.weights_to_state_dict(cfg, flat).items()})
    m = m.to(dev).train()
    tr = FusedTrainer(m, lr=1e-4)
    U = cfg.upsampling_factor
    rs = np.random.RandomState(0)
    utts = []
    for i in range(24):
        nf = int(rs.randint(600, 1200))
        utts.append((rs.uniform(-1, 1, nf * U + 5).astype(np.float32), synth.make_features(nf, 400 + i, 45.0, 300.0)))
    mean, scale = synth.scaler_stats()
    np.random.seed(1)
    gen = loaders.train_generator(utts, cfg.receptiveCausal_field, cfg.receptiveF_field, cfg.receptiveA_field, 22050,
                                  wav_transform=loaders.mu_law_transform(cfg.n_quantize), feat_transform=lambda h: (h - mean) / scale,
                                  batch_length=20000, max_length=30000, upsampling_factor=U, shuffle=True)
    stage = PinnedStager(dev)
    tgen = [0.0, 0.0, 0]

    def batches():
        it = iter(gen)
        while True:
            t0 = time.perf_counter()
            bx, bh, bt, bd, bb = next(it)
            t1 = time.perf_counter()
            dv = stage({"x": bx, "h": bh, "t": bt, "d": bd})
            t2 = time.perf_counter()
            tgen[0] += t1 - t0; tgen[1] += t2 - t1; tgen[2] += 1
            yield dv["x"], dv["h"], dv["t"], dv["d"], bb, int(np.ceil(float(bd.max())))
    for mode in ("lagged", True, False):
        stream = Prefetcher(batches())
        for _ in range(5):
            bx, bh, bt, bd, bb, maxd = next(stream)
            tr.step(bx, bh, bt, bd, bb, want_loss=mode, maxd=maxd)
        if mode == "lagged":
            tr.flush_loss()
        torch.cuda.synchronize()
        tgen[0] = tgen[1] = 0.0; tgen[2] = 0
        w_next = w_step = 0.0
        steps = 300
        t0 = time.perf_counter()
        for i in range(steps):
            a = time.perf_counter()
            bx, bh, bt, bd, bb, maxd = next(stream)
            b = time.perf_counter()
            tr.step(bx, bh, bt, bd, bb, want_loss=mode, maxd=maxd)
            c = time.perf_counter()
            w_next += b - a; w_step += c - b
        if mode == "lagged":
            tr.flush_loss()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("want_loss=%-6s: %.0f steps/s = %.0f us per iteration: main thread waits %.0f us for the batch, %.0f us in step(); loader thread per batch: generator %.0f us, stager %.0f us"
              % (mode, steps / dt, dt / steps * 1e6, w_next / steps * 1e6, w_step / steps * 1e6, tgen[0] / max(tgen[2], 1) * 1e6, tgen[1] / max(tgen[2], 1) * 1e6), flush=True)


if __name__ == "__main__":
    main()
