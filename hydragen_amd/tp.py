"""
Single-node tensor parallelism of the attention block, as /root/reference/hydragen/tp.py:30-132:
Megatron-style head sharding (q/k/v column-parallel = split heads, o_proj row-parallel) and the
all-reduce(sum) of the block output (tp.py:108-112).  One process per GPU; the collective is
`torch.distributed` all_reduce -- backend "nccl" is RCCL over xGMI on ROCm ("gloo" in CPU tests).

The first half of this file is the hot-path part (how heads / caches are partitioned and the reduce);
the second half is the caller side of SURVEY 8(f) rank 4: `apply_tp` on the model shell (tp.py:30-132),
the `{rank}.pt` shard format of make_tp_files.py:12-38 and `from_pretrained_tp` (tp.py:135-180).
"""

from __future__ import annotations

import json
from copy import deepcopy
from dataclasses import asdict
from pathlib import Path
from typing import Optional, Union

import torch
import torch.distributed as dist
from torch import Tensor, nn

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


_xgmi = None  # optional hydragen_amd.xgmi_allreduce.XgmiAllReduce


def use_xgmi_allreduce(comm) -> None:
    """Route `all_reduce_sum` through the direct xGMI all-reduce (`hyd_allreduce_sum`) instead of RCCL for tensors that
    fit its blocks; `None` switches back."""
    global _xgmi
    _xgmi = comm


def check_collectives() -> None:
    """Raise if the direct xGMI all-reduce ever gave up waiting for a peer (RCCL would have waited; the kernel's waits are
    bounded so that a lost peer cannot hang the device, and a rank that gives up publishes garbage).  Called at the end
    of every `generate()`; synchronises."""
    if _xgmi is not None:
        _xgmi.check()


def all_reduce_sum(x: Tensor) -> Tensor:
    """The forward hook of tp.py:108-112 / 83-87.  In place; no-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if _xgmi is not None and x.is_cuda and x.is_contiguous() and x.numel() * x.element_size() <= _xgmi.max_bytes \
                and x.dtype in (torch.float16, torch.bfloat16, torch.float32):
            _xgmi.all_reduce_(x)
        else:
            dist.all_reduce(x, op=dist.ReduceOp.SUM)
    return x


# ---------------------------------------------------------------------------------------------------
# Model-level tensor parallelism (tp.py:30-180, make_tp_files.py)
# ---------------------------------------------------------------------------------------------------
def _apply_tp_linear(linear: nn.Linear, style: str, rank: int, world_size: int) -> None:
    """tp.py:30-70: "colwise" splits out_features (weight dim 0, and the bias), "rowwise" splits in_features
    (weight dim 1; the bias, if any, is kept on rank 0 only so that the all-reduce adds it once)."""
    dim, attr = {"colwise": (0, "out_features"), "rowwise": (1, "in_features")}[style]
    assert getattr(linear, attr) % world_size == 0, f"{attr}={getattr(linear, attr)} does not divide over {world_size} ranks"
    w = torch.tensor_split(linear.weight, world_size, dim=dim)[rank].clone()
    linear.weight = nn.Parameter(w, requires_grad=False)
    if linear.bias is not None:
        if style == "colwise":
            linear.bias = nn.Parameter(torch.tensor_split(linear.bias, world_size, dim=0)[rank].clone(), requires_grad=False)
        elif rank != 0:
            linear.bias = nn.Parameter(torch.zeros_like(linear.bias), requires_grad=False)
    setattr(linear, attr, getattr(linear, attr) // world_size)


def _apply_tp_ffn(mlp, rank: int, world_size: int) -> None:
    """tp.py:73-87: gate/up column-parallel, down row-parallel, all-reduce(sum) of the block output."""
    _apply_tp_linear(mlp.gate_proj, "colwise", rank, world_size)
    _apply_tp_linear(mlp.up_proj, "colwise", rank, world_size)
    _apply_tp_linear(mlp.down_proj, "rowwise", rank, world_size)
    mlp.tp_reduce = True


def _apply_tp_attn(attn, rank: int, world_size: int) -> None:
    """tp.py:90-112: q/k/v column-parallel (= whole heads per rank), o_proj row-parallel, all-reduce(sum)."""
    assert attn.num_key_value_heads % world_size == 0, "kv heads must divide over the ranks (tp.py:43-46)"
    _apply_tp_linear(attn.q_proj, "colwise", rank, world_size)
    _apply_tp_linear(attn.k_proj, "colwise", rank, world_size)
    _apply_tp_linear(attn.v_proj, "colwise", rank, world_size)
    _apply_tp_linear(attn.o_proj, "rowwise", rank, world_size)
    attn.hidden_size //= world_size
    attn.num_heads //= world_size
    attn.head_dim = attn.hidden_size // attn.num_heads
    attn.num_key_value_heads //= world_size
    attn.tp_reduce = True


def apply_tp(model, rank: Optional[int] = None, world_size: Optional[int] = None) -> None:
    """Shard a `HydragenLlamaModel` (or the `.model` of a `HydragenLlamaForCausalLM`) in place for this rank
    (tp.py:115-132).  The config is overwritten BEFORE `setup_caches`, so every rank allocates caches for its own
    Hkv / N kv heads; embedding, norms and lm_head stay replicated."""
    rank = get_rank() if rank is None else rank
    world_size = get_world_size() if world_size is None else world_size
    model = getattr(model, "model", model)
    if world_size == 1:
        return
    c = model.config
    assert c.num_attention_heads % world_size == 0 and c.num_key_value_heads % world_size == 0
    c.num_attention_heads //= world_size
    c.num_key_value_heads //= world_size
    c.hidden_size //= world_size  # keeps hidden_size // num_attention_heads = head_dim (tp.py:122-124)
    for block in model.layers:
        _apply_tp_ffn(block.mlp, rank, world_size)
        _apply_tp_attn(block.self_attn, rank, world_size)


def make_tp_files(model, outdir: Union[Path, str], num_splits: int = 8) -> None:
    """make_tp_files.py:12-38: one `{rank}.pt` state dict per rank, holding that rank's shard of every
    parallel layer and a full copy of the replicated ones (+ `config.json`, the unsharded architecture,
    because there is no model hub to read it from offline)."""
    outdir = Path(outdir)
    outdir.mkdir(exist_ok=True, parents=True)
    (outdir / "config.json").write_text(json.dumps(asdict(model.config)))
    for i in range(num_splits):
        split = deepcopy(model)
        apply_tp(split.model, rank=i, world_size=num_splits)
        torch.save(split.state_dict(), outdir / f"{i}.pt")


def from_pretrained_tp(config_or_dir, load_dir: Union[Path, str, None] = None, dtype: Optional[torch.dtype] = None,
                       device: Union[str, torch.device, None] = None):
    """tp.py:135-180: build the sharded architecture without materialising full-size weights, then load
    this rank's `{rank}.pt`.  `config_or_dir` is a `LlamaConfig` or a directory holding `config.json`
    (the reference takes a hub model name here)."""
    from .llama import HydragenLlamaForCausalLM, LlamaConfig

    if load_dir is None:
        load_dir = config_or_dir
    load_dir = Path(load_dir)
    if isinstance(config_or_dir, LlamaConfig):
        config = deepcopy(config_or_dir)
    else:
        config = LlamaConfig(**json.loads((Path(config_or_dir) / "config.json").read_text()))
    world_size, rank = get_world_size(), get_rank()
    if device is None:
        device = f"cuda:{rank}" if torch.cuda.is_available() else "cpu"
    with torch.device("meta"):
        model = HydragenLlamaForCausalLM(config)
    apply_tp(model.model, rank=rank, world_size=world_size)
    part_files = sorted(load_dir.glob("*.pt"), key=lambda f: int(f.stem))
    assert len(part_files) == world_size, f"{len(part_files)} != {world_size}"
    sd = torch.load(part_files[rank], map_location=device, weights_only=True)
    model.load_state_dict(sd, assign=True)
    # non-persistent buffers (rotary tables) are not in the state dict: rebuild them on the target device
    from .llama import RotaryTable

    head_dim = config.hidden_size // config.num_attention_heads
    model.model.rotary_emb = RotaryTable(head_dim, config.max_position_embeddings, config.rope_theta, device=device)
    for layer in model.model.layers:
        layer.self_attn.rotary_emb = model.model.rotary_emb
    if dtype is None or dtype == "auto":
        dtype = next(model.parameters()).dtype
    else:
        for prm in model.parameters():
            prm.data = prm.data.to(dtype)
    model.device, model.dtype = torch.device(device), dtype
    torch.manual_seed(1234)  # make sampling consistent across ranks (tp.py:178)
    return model
