"""
The small part of /root/reference/hydragen/utils.py that sits next to the hot path:
the parity metric (`rdiff`, utils.py:13-15) and the single-node tensor-parallel bootstrap
(`get_rank`/`get_world_size`/`maybe_init_dist`, utils.py:87-133).
"""

from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def rdiff(a, b, eps=1e-8):
    diff = (a - b).abs()
    return 2 * diff / (a.abs() + b.abs() + eps)


dtype_map = {
    "float16": torch.float16,
    "bfloat16": torch.bfloat16,
    "float32": torch.float32,
}


def get_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def set_rank(rank: int):
    os.environ["LOCAL_RANK"] = str(rank)


def is_local():
    return get_rank() == 0


def local_print(*args, **kwargs):
    if is_local():
        print(*args, **kwargs)


def get_world_size() -> int:
    return int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))


def set_world_size(world_size: int):
    os.environ["LOCAL_WORLD_SIZE"] = str(world_size)


def maybe_init_dist(backend: Optional[str] = None) -> Optional[int]:
    """One process per GPU of one node (utils.py:118-133).  backend defaults to "nccl"
    (= RCCL over xGMI on ROCm) when a GPU is visible, else "gloo" (CPU tests)."""
    rank = get_rank()
    world_size = get_world_size()
    if world_size < 2:
        return None
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    if backend == "nccl":
        torch.cuda.set_device(rank)
    return rank
