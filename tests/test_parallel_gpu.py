"""GPU parity of the data-parallel compute path -- the kernels every rank runs at N > 1:
qpn_train_backward_ex (gradient weighted by the rank's row count n_r, n_r appended as the buffer's trailer) and
qpn_adam_step_ex (divides by the summed row count it reads from the trailer ON THE DEVICE), through
FusedTrainer(world_size > 1) exactly as run_train / bench.py --gpus N drive them (qpnet_amd/train.py).

Replaces the reference's dead DataParallel wrapper (src/bin/qpnet_train.py:416-423); semantics = SURVEY.md §8e:
the update is the gradient of the mean CE over ALL ranks' rows even when batch_length differs across ranks.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from qpnet_amd import synth
from qpnet_amd.config import TINY, PAPER
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _to(dev, *arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


@pytest.mark.parametrize("cfgname", ["tiny", "paper"])
def test_world2_without_exchange_equals_world1(cfgname, cuda):
    """No process group: the exchange is the identity, so the buffer holds n_r * g_r | n_r and Adam divides by n_r.
    world_size=2 must then reproduce world_size=1 step for step (loss identical, weights to fp32 rounding of
    (n*g)/n) -- a wrong trailer offset, a denominator read from the wrong word or a missing scale shows at once."""
    from qpnet_amd.train import FusedTrainer
    cfg = TINY if cfgname == "tiny" else PAPER
    bl = 600 if cfgname == "tiny" else 1500
    flat = synth.make_weights(cfg, 5)
    m1 = util.build_model(cfg, flat, cuda).train()
    m2 = util.build_model(cfg, flat, cuda).train()
    t1 = FusedTrainer(m1, lr=1e-4, world_size=1)
    t2 = FusedTrainer(m2, lr=1e-4, world_size=2)
    for step in range(3):
        x, h, t, d, b = synth.train_inputs(cfg, bl + 37 * step, 70 + step, 30000)
        xt, ht, tt, dt, bt = _to(cuda, x, h, t, d, b)
        if step:
            # both trainers step from the SAME weights: after a step they differ by the rounding of (n g) / n (checked below), and a forward from weights
            # 1e-7 apart may put a post-net pre-activation on the other side of a ReLU kink -- a gradient difference that is no property of the exchange
            assert float((m2.flat_parameters() - m1.flat_parameters()).abs().max()) <= 2e-7
            with __import__("torch").no_grad():
                m2._flat.copy_(m1._flat)                      # (the flat buffer the parameters are views of: train.ensure_flat)
        l1 = t1.step(xt, ht, tt, dt, bt)
        l2 = t2.step(xt, ht, tt, dt, bt)
        assert abs(l1 - l2) < 1e-6          # (the loss is summed with double atomics: order differs run to run)
        BL = int(b[0])
        n = m1.flat_parameters().numel()
        g1 = t1.g[:n].cpu().numpy().astype(np.float64)
        g2 = t2.g[:n].cpu().numpy().astype(np.float64)
        trailer = t2.g[n:].cpu().numpy()
        np.testing.assert_array_equal(trailer, np.array([x.shape[0] * BL, 0, 0, 0], np.float32))
        # the weighted gradient is n_r * g_r (the adaptive blocks' scatter-add uses float atomics, so two runs of the same
        # backward differ by reassociation: tolerance relative to the largest gradient)
        nrow = x.shape[0] * BL
        np.testing.assert_allclose(g2, g1 * nrow, rtol=0, atol=2e-6 * np.abs(g1).max() * nrow)
        # Adam's update is invariant to a constant scale of g, its first moment is not: m = (1 - b1) * g / denominator pins
        # the division by the trailer's row count inside qpn_adam_step_ex
        mm1 = t1.m.cpu().numpy().astype(np.float64); mm2 = t2.m.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(mm2, mm1, rtol=0, atol=2e-6 * np.abs(mm1).max())
        vv1 = t1.v.cpu().numpy().astype(np.float64); vv2 = t2.v.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(vv2, vv1, rtol=0, atol=4e-6 * np.abs(vv1).max())
    w1 = m1.flat_parameters().cpu().numpy()
    w2 = m2.flat_parameters().cpu().numpy()
    np.testing.assert_allclose(w2, w1, atol=2e-7, rtol=0)
    assert np.abs(w1 - flat).max() > 1e-5          # the three steps did move the weights


_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
rank = int(sys.argv[1]); world = int(sys.argv[2]); out = sys.argv[3]
os.environ["MASTER_ADDR"] = "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
from qpnet_amd import synth, parallel
from qpnet_amd.config import TINY, PAPER
from qpnet_amd.train import FusedTrainer, ensure_flat
import util
cfg = {{"tiny": TINY, "paper": PAPER}}[sys.argv[4]]
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
m = util.build_model(cfg, synth.make_weights(cfg, 3 + rank), dev).train()      # deliberately different: broadcast must fix it
parallel.broadcast_parameters(ensure_flat(m, dev), 0)
tr = FusedTrainer(m, lr=1e-4, world_size=world)
bls = [int(v) for v in sys.argv[5].split(",")]                                  # unequal batch_length across ranks
losses = []
for ci in parallel.shard_indices(4, rank, world):
    x, h, t, d, b = synth.train_inputs(cfg, bls[ci], 900 + ci, 30000)
    xs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, h, t, d, b)]
    losses.append(tr.step(*xs))
np.savez(out, w=m.flat_parameters().cpu().numpy(), losses=np.array(losses), buckets=np.array(tr.last_buckets, dtype=np.int64))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("cfgname,buckets", [("tiny", 1), ("tiny", 2), ("paper", 2), ("tiny", "mixed")])
def test_two_ranks_on_one_gpu_equal_union_batch_oracle(cfgname, buckets, cuda, tmp_path):
    """Two fresh child processes share GPU 0 and exchange over gloo (RCCL refuses two ranks on one device); each runs the
    product's FusedTrainer(world_size=2) on chunks of unequal batch_length.  Final weights must be bit-identical across
    the ranks and equal the numpy oracle trained on the UNION batch of every step (row-weighted mean).
    [paper: BASELINE config[2]'s geometry; its forward is the one-launch residual stack (csrc/train_stack.hip), whose two
    processes' persistent workgroups share the card here -- positions are handed out by tickets, so neither needs all of its
    workgroups resident to finish.]"""
    from oracle import train_oracle as TO
    port = 29500 + os.getpid() % 2000
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT))
    # "mixed": the ranks' environments DISAGREE (rank 0 asks for two buckets, rank 1 for one): the split is agreed over the process group at the
    # first step (train.FusedTrainer._agree_two_buckets), so both fall back to ONE exchange instead of issuing different numbers of collectives
    envs = [dict(os.environ, MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", QPN_EXCHANGE_BUCKETS=str((2, 1)[r] if buckets == "mixed" else buckets)) for r in range(2)]
    outs = [str(tmp_path / ("r%d.npz" % r)) for r in range(2)]
    cfg = TINY if cfgname == "tiny" else PAPER
    bls = [300, 410, 350, 280] if cfgname == "tiny" else [1400, 1750, 1500, 1250]
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", outs[r], cfgname, ",".join(map(str, bls))], env=envs[r]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=420) == 0
    r0, r1 = np.load(outs[0]), np.load(outs[1])
    np.testing.assert_array_equal(r0["w"], r1["w"])                 # replicas stay bit-identical
    # QPN_EXCHANGE_BUCKETS=2: the exchange went out as two buckets: the post-net blocks + trailer (final while the layer backward still runs) first, the rest behind them
    n_post = cfg.n_skipch * cfg.n_skipch + cfg.n_skipch + cfg.n_quantize * cfg.n_skipch + cfg.n_quantize
    for r in (r0, r1):
        if buckets == 2:
            assert int(r["buckets"][1]) == n_post + 4 and int(r["buckets"][0]) + n_post == r0["w"].size
        else:
            assert int(r["buckets"][1]) == 0
    flat = synth.make_weights(cfg, 3)
    opt = TO.Adam(flat.size)
    for step in range(2):
        gs, ns, ls = [], [], []
        for ci in (2 * step, 2 * step + 1):
            x, h, t, d, b = synth.train_inputs(cfg, bls[ci], 900 + ci, 30000)
            lg, caches = TO.forward(cfg, flat, x, h, d, b)
            BL = int(b[0])
            loss, dl = TO.ce_loss(lg, t[:, -BL:])
            gs.append(TO.backward(cfg, flat, caches, dl)); ns.append(BL); ls.append(loss)
        assert abs(r0["losses"][step] - ls[0]) < 1e-4 and abs(r1["losses"][step] - ls[1]) < 1e-4     # north_star tolerance
        g = (gs[0] * ns[0] + gs[1] * ns[1]) / (ns[0] + ns[1])
        opt.step(flat, g.astype(np.float32))
    # (Adam divides by sqrt(v) + 1e-8: where a gradient is ~1e-7 the update moves by 80x the gradient's absolute error, so the order in which
    #  the kernels sum their partial slabs shows at the 1e-6 level -- one element of 504 495 at 2.09e-6 with 48 instead of 64 slabs;
    #  north_star's tolerance, 1e-4 on the losses, is the assert above)
    np.testing.assert_allclose(r0["w"], flat, atol=4e-6, rtol=0)


_FLAG_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
rank = int(sys.argv[1]); world = int(sys.argv[2]); out = sys.argv[3]
os.environ["MASTER_ADDR"] = "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
from qpnet_amd import synth, parallel, _lib
from qpnet_amd.config import TINY
from qpnet_amd.train import FusedTrainer, ensure_flat
import util
cfg = TINY
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
m = util.build_model(cfg, synth.make_weights(cfg, 3), dev).train()
parallel.broadcast_parameters(ensure_flat(m, dev), 0)
tr = FusedTrainer(m, lr=1e-3, world_size=world)
ws, codes, msgs = [], [], []
for step in range(3):
    x, h, t, d, b = synth.train_inputs(cfg, 300 + 40 * rank, 900 + 2 * step + rank, 30000)
    if step == 1 and rank == 1:
        t = t.copy(); t[0, -7] = cfg.n_quantize + 5                      # rank 1's chunk of step 1 is bad; rank 0's is clean
    xs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, h, t, d, b)]
    try:
        tr.step(*xs)                                                     # (want_loss=True: the status is read in the step, like the reference's asserts)
        codes.append(0); msgs.append("")
    except _lib.QpnError as e:
        codes.append(e.code); msgs.append(str(e))
    ws.append(m.flat_parameters().cpu().numpy().copy())
np.savez(out, w=np.stack(ws), codes=np.array(codes), steps=np.array([tr.step_count]), peer=np.array(["peer rank" in s for s in msgs]))
dist.barrier()
dist.destroy_process_group()
"""


def test_a_step_flagged_on_one_rank_is_skipped_by_every_rank(cuda, tmp_path):
    """the Adam kernel skips a step whose device-side status word is set -- but on the flagged rank only, while its peers applied the summed
    gradient: the replicas diverged.  The flag now rides in the exchanged trailer (word 1, summed like the row count), every rank's Adam kernel skips the
    step, the flagged rank raises its own error and the others "a peer rank flagged ...": weights bit-identical across the ranks after every step, unchanged
    by the flagged one, the step counts set back to the two updates applied."""
    port = 29500 + (os.getpid() + 11) % 2000
    script = tmp_path / "flag.py"
    script.write_text(_FLAG_SCRIPT.format(root=ROOT))
    env = dict(os.environ, MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / ("f%d.npz" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", outs[r]], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=420) == 0
    r0, r1 = np.load(outs[0]), np.load(outs[1])
    np.testing.assert_array_equal(r0["w"], r1["w"])                              # after every step, flagged or not
    assert list(r0["codes"]) == [0, -4, 0] and list(r1["codes"]) == [0, -4, 0]
    assert bool(r0["peer"][1]) and not bool(r1["peer"][1])                       # rank 0 learns it from the trailer, rank 1 from its own word
    np.testing.assert_array_equal(r0["w"][1], r0["w"][0])                        # the flagged step moved nothing ...
    assert np.abs(r0["w"][2] - r0["w"][1]).max() > 1e-4                          # ... and training goes on
    assert int(r0["steps"][0]) == 2 and int(r1["steps"][0]) == 2
