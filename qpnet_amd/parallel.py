"""Data-parallel training helpers: utterance-chunk sharding and the one gradient exchange per step.

The reference's only multi-GPU code is a dead `torch.nn.DataParallel` wrapper (src/bin/qpnet_train.py:416-423,
batch_size 1 / n_gpus 1).  Here: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
MI355X, "gloo" in CPU tests), rank r consumes chunks r, r+N, r+2N, ... of the generator stream, and the flat
fp32 gradient (2.0 MB for the paper-size model) is all-reduced ONCE per step - no bucketing/overlap needed at
this size.  The reference's loss is a mean over one batch with a common batch_length
(qpnet.py:253); across ranks batch_length may differ (it depends on max d in each rank's buffer,
qpnet_train.py:268-284), so gradients are weighted by each rank's row count to reproduce the GLOBAL mean.
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """chunk ids consumed by `rank`: r, r+N, r+2N, ... (round-robin keeps ranks in lock-step)."""
    return list(range(rank, n_items, world_size))


def allreduce_mean_gradient(gflat, n_rows_local, group=None):
    """In place: g <- sum_r(n_r * g_r) / sum_r(n_r), i.e. the gradient of the mean CE over ALL ranks' rows.

    `gflat` is the rank's gradient of ITS mean loss (what qpn_train_backward returns), `n_rows_local` = B*BL.
    One all-reduce of n_params+1 floats."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return gflat
    buf = torch.empty(gflat.numel() + 1, dtype=gflat.dtype, device=gflat.device)
    buf[:-1] = gflat * float(n_rows_local)
    buf[-1] = float(n_rows_local)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    gflat.copy_(buf[:-1] / buf[-1])
    return gflat


def broadcast_parameters(flat, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (reference: single process, one model)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat
