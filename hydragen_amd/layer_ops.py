"""Python face of `hyd_add_rmsnorm` / `hyd_swiglu` (include/hydragen_hip.h): the elementwise glue of the decoder layer
around the attention block -- residual add + RMSNorm (/root/reference/hydragen/llama.py:615-631 with transformers'
LlamaRMSNorm, llama.py:605-608,656) and the SwiGLU gate (transformers' LlamaMLP, llama.py:2,604) -- each as one
HIP kernel instead of two torch launches."""

from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import AddRmsnormParams, SwigluParams
from .flash import _dtype_code, _require_gpu, _stream


def supported(x: Tensor, n_max: int = 16384) -> bool:
    """Shapes the kernels take: 16-bit CUDA rows, contiguous in the last dimension, a multiple of 8 wide, every row
    16-byte aligned (what the C entry points check); anything else takes the torch form in the model shell."""
    return (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.shape[-1] % 8 == 0 and x.shape[-1] <= n_max
            and x.stride(-1) == 1 and x.data_ptr() % 16 == 0 and all(st % 8 == 0 for st in x.stride()[:-1]))


def _rows(t: Tensor) -> Tensor:
    """[..., n] -> [rows, n] view with one row stride (no copy for the views the model shell produces)."""
    return t.reshape(-1, t.shape[-1])


def add_rms_norm(x: Tensor, residual: Optional[Tensor], weight: Tensor, eps: float):
    """(residual + x, RMSNorm(residual + x) * weight); residual None: (x, RMSNorm(x) * weight).  The sum is rounded to
    x.dtype before the statistic is taken (it is the stored residual stream); the norm is fp32 inside, one rounding."""
    _require_gpu(x, weight)
    lib = _lib.load()
    x2 = _rows(x)
    n = x2.shape[1]
    assert x2.stride(1) == 1 and n % 8 == 0 and weight.shape == (n,) and weight.dtype == x.dtype and weight.is_contiguous()
    normed = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if x2.shape[0] == 0:
        return (x if residual is None else torch.empty_like(normed)), normed
    p = AddRmsnormParams()
    p.x, p.weight, p.norm_out = x2.data_ptr(), weight.data_ptr(), normed.data_ptr()
    p.x_row_stride, p.norm_row_stride = x2.stride(0) if x2.shape[0] > 1 else n, n
    summed = x
    if residual is not None:
        _require_gpu(residual)
        r2 = _rows(residual)
        assert r2.shape == x2.shape and r2.dtype == x.dtype and r2.stride(1) == 1
        summed = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        p.residual, p.sum_out = r2.data_ptr(), summed.data_ptr()
        p.residual_row_stride, p.sum_row_stride = r2.stride(0) if r2.shape[0] > 1 else n, n
    p.rows, p.n, p.dtype, p.eps = x2.shape[0], n, _dtype_code(x), float(eps)
    _lib.check(lib.hyd_add_rmsnorm(C.byref(p), _stream()))
    return summed, normed


def swiglu(gate: Tensor, up: Tensor) -> Tensor:
    """silu(gate) * up; gate / up [..., n] may be the column halves of one fused GEMM output (row-strided views)."""
    _require_gpu(gate, up)
    lib = _lib.load()
    g2, u2 = _rows(gate), _rows(up)
    n = g2.shape[1]
    assert g2.shape == u2.shape and gate.dtype == up.dtype and g2.stride(1) == 1 and u2.stride(1) == 1 and n % 8 == 0
    out = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
    if g2.shape[0] == 0:
        return out
    p = SwigluParams()
    p.gate, p.up, p.out = g2.data_ptr(), u2.data_ptr(), out.data_ptr()
    one = g2.shape[0] <= 1
    p.gate_row_stride, p.up_row_stride, p.out_row_stride = (n if one else g2.stride(0)), (n if one else u2.stride(0)), n
    p.rows, p.n, p.dtype = g2.shape[0], n, _dtype_code(gate)
    _lib.check(lib.hyd_swiglu(C.byref(p), _stream()))
    return out


_HYD_F32 = 2


def _next_sample_key(device: torch.device):
    """(seed, offset) of the next sampling call, taken from -- and advancing -- torch's CUDA generator of the device: the
    draw is a function of torch.manual_seed and of the random ops issued since, like torch.multinomial's (the reference
    seeds every tensor-parallel rank alike so that all ranks draw the same tokens, tp.py:178)."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed, offset = gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, gen.get_offset()
    gen.set_offset(offset + 4)
    return seed, offset


def sample_tokens(logits: Tensor, temperature: float, key: Optional[tuple] = None) -> Tensor:
    """[B, V] logits (fp16 / bf16 / fp32, rows contiguous) -> [B, 1] int64 tokens drawn from softmax(logits / temperature)
    (temperature 0: argmax) in one kernel."""
    _require_gpu(logits)
    lib = _lib.load()
    assert logits.ndim == 2 and logits.stride(1) == 1 and logits.shape[1] > 0
    out = torch.empty((logits.shape[0], 1), dtype=torch.int64, device=logits.device)
    if logits.shape[0] == 0:
        return out
    seed, offset = key if key is not None else _next_sample_key(logits.device)
    p = _lib.SampleParams()
    p.logits, p.out, p.row_stride = logits.data_ptr(), out.data_ptr(), logits.stride(0) if logits.shape[0] > 1 else logits.shape[1]
    p.seed, p.offset, p.rows, p.n = seed, offset, logits.shape[0], logits.shape[1]
    p.dtype = _HYD_F32 if logits.dtype == torch.float32 else _dtype_code(logits)
    p.temperature = float(temperature)
    _lib.check(lib.hyd_sample_tokens(C.byref(p), _stream()))
    return out
