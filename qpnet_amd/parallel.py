"""Data-parallel training helpers: utterance-chunk sharding and the one gradient exchange per step.

The reference's only multi-GPU code is a dead `torch.nn.DataParallel` wrapper (src/bin/qpnet_train.py:416-423,
batch_size 1 / n_gpus 1).  Here: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
MI355X, "gloo" in CPU tests), rank r consumes chunks r, r+N, r+2N, ... of the generator stream, and the flat
fp32 gradient (2.0 MB for the paper-size model) is all-reduced ONCE per step; QPN_EXCHANGE_BUCKETS=2 sends the tail the
device finishes early (post-net blocks + trailer, a quarter of the buffer) as its own bucket under the layer backward (exchange_two_buckets).  The reference's loss is a mean over one batch with a common batch_length
(qpnet.py:253); across ranks batch_length may differ (it depends on max d in each rank's buffer,
qpnet_train.py:268-284), so gradients are weighted by each rank's row count to reproduce the GLOBAL mean.
"""
import os

import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """chunk ids consumed by `rank`: r, r+N, r+2N, ... (round-robin keeps ranks in lock-step)."""
    return list(range(rank, n_items, world_size))


TRAILER = 4      # floats behind the gradient in the exchange buffer: {row count, 0, 0, 0} (keeps the payload 16-byte sized)


def exchange(buf, group=None):
    """THE gradient exchange of a step: one all-reduce(SUM) of the flat buffer [n_r * g_r | n_r, 0, 0, 0] (RCCL over xGMI on
    MI355X; 2.0 MB for the paper-size model, so one unbucketed call).  In the fused step the HIP kernels produce the weighted
    buffer (qpn_train_backward_ex) and consume the sum (qpn_adam_step_ex divides by the summed row count on the device)."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("QPN_EXCHANGE_ALWAYS")):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)      # QPN_EXCHANGE_ALWAYS: one-rank rehearsal of the RCCL call
    return buf


def exchange_two_buckets(buf, first, count, group=None, early_stream=None):
    """The same exchange as two all-reduces: buf[first : first + count] -- what the device finishes while the layer backward is still running
    (the post-net gradient blocks and the trailer: qpn_train_early_bucket) -- goes first, on `early_stream` (a torch.cuda.Stream that already
    waits for exactly that range; None on the CPU), and buf[:first] follows on the current stream, which is joined with `early_stream`
    afterwards.  At 8 ranks the early quarter of the 2 MB buffer is then off the step's critical path.  Element for element the result is
    that of exchange(buf): an all-reduce sums each element over the ranks independently of its neighbours.  count == 0: one exchange."""
    n = buf.numel()
    if count <= 0 or first <= 0 or first + count != n:
        return exchange(buf, group)
    if early_stream is not None:
        with torch.cuda.stream(early_stream):
            exchange(buf[first:], group)
        exchange(buf[:first], group)
        torch.cuda.current_stream(buf.device).wait_stream(early_stream)
    else:
        exchange(buf[first:], group)
        exchange(buf[:first], group)
    return buf


def allreduce_mean_gradient(gflat, n_rows_local, group=None, early_first=None):
    """In place: g <- sum_r(n_r * g_r) / sum_r(n_r), i.e. the gradient of the mean CE over ALL ranks' rows, for callers
    that hold a plain gradient tensor (the reference-style autograd loop).  Same exchange, weighting done with torch ops.
    early_first: first element of the tail that goes out as its own bucket (exchange_two_buckets), None: one exchange."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return gflat
    buf = torch.zeros(gflat.numel() + TRAILER, dtype=gflat.dtype, device=gflat.device)
    buf[:gflat.numel()] = gflat * float(n_rows_local)
    buf[gflat.numel()] = float(n_rows_local)
    if early_first is None:
        exchange(buf, group)
    else:
        exchange_two_buckets(buf, int(early_first), buf.numel() - int(early_first), group)
    gflat.copy_(buf[:gflat.numel()] / buf[gflat.numel()])
    return gflat


def broadcast_parameters(flat, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (reference: single process, one model)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def replica_drift(flat, src=0, group=None):
    """max |w_r - w_src| over the ranks (0.0 on every rank when the replicas are bit-identical, as identical Adam steps on an identical summed
    gradient keep them).  One broadcast + one all-reduce of the parameter vector: a check for the END of a run, not for the step."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0.0
    ref = flat.detach().clone()
    dist.broadcast(ref, src=src, group=group)
    d = (flat.detach() - ref).abs().max().reshape(1)
    dist.all_reduce(d, op=dist.ReduceOp.MAX, group=group)
    return float(d.item())
