"""
Single-node tensor parallelism of the attention block, as /root/reference/hydragen/tp.py:30-132:
Megatron-style head sharding (q/k/v column-parallel = split heads, o_proj row-parallel) and the
all-reduce(sum) of the block output (tp.py:108-112).  One process per GPU; the collective is
`torch.distributed` all_reduce -- backend "nccl" is RCCL over xGMI on ROCm ("gloo" in CPU tests).

Only the pieces on the hot path live here: how heads/caches are partitioned and the reduce.
Weight-file sharding/loading (tp.py:135-180, make_tp_files.py) is out of scope.
"""

from __future__ import annotations

import torch
import torch.distributed as dist
from torch import Tensor

from .utils import get_rank, get_world_size


def shard_range(n: int, rank: int | None = None, world_size: int | None = None) -> slice:
    """Contiguous slice of `n` heads owned by `rank` (torch.tensor_split semantics of tp.py:49-50
    for the evenly divisible case the reference asserts, tp.py:43-46)."""
    rank = get_rank() if rank is None else rank
    world_size = get_world_size() if world_size is None else world_size
    assert n % world_size == 0, f"{n} heads do not divide over {world_size} ranks"
    per = n // world_size
    return slice(rank * per, (rank + 1) * per)


def shard_heads(x: Tensor, head_dim_index: int = -2, rank: int | None = None, world_size: int | None = None) -> Tensor:
    """Slice the head axis of q / k / v / a shared-cache level for this rank (tp.py:103-106,121-123)."""
    sl = shard_range(x.shape[head_dim_index], rank, world_size)
    idx = [slice(None)] * x.ndim
    idx[head_dim_index] = sl
    return x[tuple(idx)].contiguous()


def shard_attention_inputs(q, k, v, shared_ks, shared_vs, rank=None, world_size=None):
    """Per-rank inputs of `hydragen_attention`: every tensor keeps its layout, only the head axis
    shrinks (Hq/N query heads, Hkv/N kv heads), so each rank runs the identical operator."""
    f = lambda t: shard_heads(t, -2, rank, world_size)
    return f(q), f(k), f(v), [f(t) for t in shared_ks], [f(t) for t in shared_vs]


def shard_o_proj_weight(w: Tensor, rank=None, world_size=None) -> Tensor:
    """Row-parallel o_proj (tp.py:96-101): weight [hidden, Hq*D] split along in_features."""
    sl = shard_range(w.shape[1], rank, world_size)
    return w[:, sl].contiguous()


def all_reduce_sum(x: Tensor) -> Tensor:
    """The forward hook of tp.py:108-112 / 83-87.  In place; no-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
    return x
