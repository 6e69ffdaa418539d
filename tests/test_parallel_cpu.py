"""CPU, world_size 2 and 8 over gloo: the data-parallel step (chunk sharding + one weighted flat-gradient all-reduce
+ identical Adam on every rank) equals single-process training on the union of the two ranks' rows.
Gradients come from the numpy oracle here (the HIP kernels need a GPU); the exchange code is the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out, two_buckets=False):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qpnet_amd import synth, parallel
    from qpnet_amd.config import TINY
    from oracle import train_oracle as TO
    cfg = TINY
    flat = synth.make_weights(cfg, 3 + rank)          # deliberately different: broadcast must fix it
    ft = torch.from_numpy(flat)
    parallel.broadcast_parameters(ft, 0)
    flat = ft.numpy().copy()
    opt = TO.Adam(flat.size)
    bls = _bls(world)                                 # unequal batch_length across ranks
    chunks = parallel.shard_indices(len(bls), rank, world)   # 2 steps of `world` ranks
    for step, ci in enumerate(chunks):
        x, h, t, d, b = synth.train_inputs(cfg, bls[ci], 900 + ci, 30000)
        lg, caches = TO.forward(cfg, flat, x, h, d, b)
        BL = int(b[0])
        loss, dl = TO.ce_loss(lg, t[:, -BL:])
        g = torch.from_numpy(TO.backward(cfg, flat, caches, dl))
        # two buckets: the post-net blocks (the tail of the flat order) + trailer first, the rest behind them -- what FusedTrainer does on the GPU
        early = flat.size - (cfg.n_skipch * cfg.n_skipch + cfg.n_skipch + cfg.n_quantize * cfg.n_skipch + cfg.n_quantize) if two_buckets else None
        parallel.allreduce_mean_gradient(g, x.shape[0] * BL, early_first=early)
        opt.step(flat, g.numpy())
    out[rank] = flat
    dist.barrier()
    dist.destroy_process_group()


def _bls(world):
    """batch_length of chunk ci (2 steps x world ranks), unequal across the ranks of a step"""
    return [300, 410, 350, 280] if world == 2 else [200 + 37 * ((5 * ci) % 11) for ci in range(2 * world)]


@pytest.mark.parametrize("world,two_buckets", [(2, False), (2, True), (8, False), (8, True)])
def test_two_rank_step_equals_global_batch(world, two_buckets):
    """world 8 = the rank count of BASELINE config[2]: the exchange, the row-count weighting and the two-bucket agreement at the target
    world size (gloo on the CPU; eight single-threaded workers)."""
    mp.set_start_method("spawn", force=True)
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + (7 if two_buckets else 0) + world
    procs = [mp.Process(target=_worker, args=(r, world, port, out, two_buckets)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
        assert p.exitcode == 0
    w0 = out[0]
    for r in range(1, world):
        np.testing.assert_array_equal(w0, out[r])     # replicas stay bit-identical
    # single-process reference: each step = mean CE over the rows of both chunks of that step
    sys.path.insert(0, ROOT)
    from qpnet_amd import synth
    from qpnet_amd.config import TINY
    from oracle import train_oracle as TO
    cfg = TINY
    flat = synth.make_weights(cfg, 3)
    opt = TO.Adam(flat.size)
    bls = _bls(world)
    for step in range(2):
        gs, ns = [], []
        for ci in range(world * step, world * (step + 1)):
            x, h, t, d, b = synth.train_inputs(cfg, bls[ci], 900 + ci, 30000)
            lg, caches = TO.forward(cfg, flat, x, h, d, b)
            BL = int(b[0])
            _, dl = TO.ce_loss(lg, t[:, -BL:])
            gs.append(TO.backward(cfg, flat, caches, dl)); ns.append(BL)
        g = sum(gi.astype(np.float64) * n for gi, n in zip(gs, ns)) / float(sum(ns))
        opt.step(flat, g.astype(np.float32))
    np.testing.assert_allclose(w0, flat, atol=2e-7 if world == 2 else 1e-6, rtol=0)


def test_shard_indices():
    from qpnet_amd.parallel import shard_indices
    assert shard_indices(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((shard_indices(10, r, 4) for r in range(4)), [])) == list(range(10))
