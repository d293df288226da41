"""Data-parallel sharding of the MS-HGNN path across the GPUs of one node.

Windows are independent, so the path shards with no data-path collective: the global minibatch is split
contiguously across ranks (one process per GPU), weights are replicated, and the ONLY exchange per step is
one sum-all-reduce of the flat fp32 gradient buffer the C-ABI produces (3.98 MB for A1-C2 h128 L3) -- RCCL
over xGMI on GPUs (`backend="nccl"`), gloo on CPU in the tests.  This is what Lightning-DDP would do for the
reference with devices>1 (gnnLightning.py:1396-1400), with mean semantics (sum / world size).

The nn.Module surface (models.py) needs none of this: its parameters are ordinary nn.Parameters, so
torch.nn.parallel.DistributedDataParallel / Lightning DDP wrap it unchanged.
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) window range of `rank`; the first (global_batch % world) ranks get one more."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(global_batch, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard_x_dict(x_dict, num_nodes, global_batch: int, rank: int, world: int):
    """Slice reference-convention inputs ([B*n_t, F_t], graph-major) to this rank's windows."""
    b, e = shard_bounds(global_batch, rank, world)
    return {t: x[b * num_nodes[t]: e * num_nodes[t]] for t, x in x_dict.items()}, (b, e)


def allreduce_gradients_(flat_grad: torch.Tensor, local_windows: int, global_windows: int, group=None) -> torch.Tensor:
    """In place: turn per-rank gradients of the LOCAL mean loss into the gradient of the GLOBAL mean loss.
    grad_global = sum_r (local_windows_r / global_windows) * grad_r  -- one all-reduce of the flat buffer."""
    import torch.distributed as dist
    flat_grad.mul_(float(local_windows) / float(global_windows))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad
