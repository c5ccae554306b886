"""
Attention primitives with the reference's signatures (/root/reference/hydragen/flash.py),
implemented by the hand-written gfx950 kernels behind the C ABI in include/hydragen_hip.h.

    flash_attention          flash.py:284-306   -> hyd_prefix_attn_fwd (MFMA kernel)
    flash_attention_varlen   flash.py:309-351   -> hyd_prefix_attn_fwd (packed K/V + cu_seqlens)
    flash_attention_seqlen   flash.py:163-281   -> hyd_suffix_attn_fwd (wavefront GEMV kernel)

PyTorch is used for device memory and the current stream only.
"""

from __future__ import annotations

import contextvars
import ctypes as C

import torch
from torch import Tensor

from . import _lib
from ._lib import HYD_BF16, HYD_F16, HYD_LSE_BHQ, HYD_LSE_BQH, PrefixParams, SuffixParams


def _dtype_code(t: Tensor) -> int:
    if t.dtype == torch.float16:
        return HYD_F16
    if t.dtype == torch.bfloat16:
        return HYD_BF16
    raise NotImplementedError(f"hydragen_amd kernels take float16/bfloat16, got {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device (the raw accessor skips ~8 us of
    Python per call; every launch of the library goes to this stream, so graph capture sees it)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(*ts: Tensor):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "hydragen_amd operators run only on a ROCm GPU through libhydragen_hip.so; "
                f"got a tensor on {t.device} (there is deliberately no CPU fallback)"
            )


def _lastdim_contig(t: Tensor) -> Tensor:
    """K/V may be arbitrary views as long as head_dim is contiguous and strides are 16B-aligned."""
    if t.is_contiguous() and t.data_ptr() % 16 == 0 and t.shape[-1] % 8 == 0:  # the common case, one C call
        return t
    if t.data_ptr() % 16:  # a view at an odd storage offset: .contiguous() would hand the same memory back
        return t.clone(memory_format=torch.contiguous_format)
    if t.stride(-1) != 1 or any(s % 8 for s in t.stride()[:-1]):
        return t.contiguous()
    return t


def _q_contig(q: Tensor) -> Tensor:
    """q / out are addressed as dense [.., Hq, D] rows with 16-byte accesses."""
    if q.data_ptr() % 16:
        return q.clone(memory_format=torch.contiguous_format)
    return q.contiguous()


# ---- head dims other than the kernels' 64 / 128 / 256 ---------------------------------------------------------------
# flash-attn takes any head_dim that is a multiple of 8 up to 256 (flash.py:295-304 hands it whatever the model has).
# The HIP kernels are instantiated for 64, 128 and 256; other multiples of 8 run zero-PADDED to the next of the three
# with the TRUE head dim's softmax scale (hyd_*_params.softmax_scale): zero columns add nothing to q.k and produce zero
# output columns, which are cut off again.  Functional, not fast: q, k and v are copied on every call -- a model with
# such a head dim should keep its caches padded instead.
# forwarded to the C ABI by the marshalling helpers below; 0 = head_dim ** -0.5.  A context variable: two threads (or
# tasks) serving models with different head dims each see their own value.
_scale_var: contextvars.ContextVar = contextvars.ContextVar("hydragen_amd_softmax_scale", default=0.0)


def current_softmax_scale() -> float:
    return _scale_var.get()


def padded_head_dim(d: int) -> int:
    if d in (64, 128, 256):
        return d
    if d <= 0 or d % 8 or d > 256:
        raise NotImplementedError(f"head_dim {d}: multiples of 8 up to 256 are implemented (as in flash-attn)")
    return 64 if d < 64 else 128 if d < 128 else 256


def pad_head_dim(t: Tensor, dp: int) -> Tensor:
    return torch.nn.functional.pad(t, (0, dp - t.shape[-1]))


class true_head_dim_scale:
    """with true_head_dim_scale(d): every call marshalled inside uses softmax scale d ** -0.5."""

    def __init__(self, d: int):
        self.scale = float(d) ** -0.5

    def __enter__(self):
        self.token = _scale_var.set(self.scale)

    def __exit__(self, *exc):
        _scale_var.reset(self.token)


# ---- schedule hint for ragged unique lengths ---------------------------------------------------------------------------
# The suffix kernels hand sequences to the chip in index order.  With ragged lengths the last workgroups of a launch are then
# a random mix, the short ones leave, and the long ones finish on a half-empty chip: C2 heads, lengths 1..128 at random, 184 us
# against 163 us for the same keys in equal rows (tests/probes/ragged_lengths_probe.py).  Handing the LONGEST sequences out
# first brings that to 169 us.  Only the caller can know the order cheaply -- during decode every length grows by one per step, so
# one argsort at the start of a generation serves all of its steps (hydragen_amd/llama.py does that) -- and the reference's
# operator signatures (attention.py:177-392, flash.py:163-281) have no argument for it, so it travels like the softmax scale
# above: a context variable that the marshalling helpers forward to hyd_suffix_params.seq_order.  None = index order.
_order_var: contextvars.ContextVar = contextvars.ContextVar("hydragen_amd_seq_order", default=None)


def current_seq_order() -> Tensor | None:
    return _order_var.get()


def longest_first(seq_lens: Tensor) -> Tensor:
    """The schedule for `seq_lens` [B]: int32 permutation, longest sequence first (stable)."""
    return torch.argsort(seq_lens, descending=True, stable=True).to(torch.int32).contiguous()


class seq_order:
    """with seq_order(perm): suffix passes marshalled inside hand sequence perm[i] to dispatch slot i.  `perm`: int32 [B] on the
    device, a permutation of 0..B-1 (checked here once, on entry: the kernels trust it), or None."""

    def __init__(self, perm: Tensor | None, check: bool = True):
        if perm is not None:
            if perm.dtype != torch.int32 or perm.ndim != 1 or not perm.is_contiguous():
                raise ValueError(f"seq_order: contiguous int32 [B] expected, got {perm.dtype} {tuple(perm.shape)}")
            if check and not torch.equal(torch.sort(perm).values, torch.arange(perm.numel(), dtype=torch.int32, device=perm.device)):
                raise ValueError("seq_order: not a permutation of 0..B-1")
        self.perm = perm

    def __enter__(self):
        self.token = _order_var.set(self.perm)
        return self

    def __exit__(self, *exc):
        _order_var.reset(self.token)


def prefix_attention(
    q: Tensor, k: Tensor, v: Tensor, *, sb: int, kv_len: int, group_stride: tuple[int, int],
    tok_stride: tuple[int, int], head_stride: tuple[int, int], B: int, nq: int, causal: bool,
    lse_layout: int, lse_shape, cu_seqlens_k: Tensor | None = None, cu_seqlens_q: Tensor | None = None,
    max_q_len: int = 0, want_lse: bool = True, num_splits: int = 0, out: Tensor | None = None,
):
    """Thin marshalling of hyd_prefix_attn_fwd; q must be contiguous [.., Hq, D]."""
    lib = _lib.load()
    Hq, D = q.shape[-2], q.shape[-1]
    Hkv = k.shape[-2]
    p = PrefixParams()
    p.q, p.k, p.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    if out is None:
        out = torch.empty_like(q)
    p.out = out.data_ptr()
    lse = None
    if want_lse:
        lse = torch.zeros(lse_shape, dtype=torch.float32, device=q.device) if cu_seqlens_q is not None else \
            torch.empty(lse_shape, dtype=torch.float32, device=q.device)
        p.lse = lse.data_ptr()
    p.cu_seqlens_k = cu_seqlens_k.data_ptr() if cu_seqlens_k is not None else None
    p.cu_seqlens_q = cu_seqlens_q.data_ptr() if cu_seqlens_q is not None else None
    p.k_group_stride, p.v_group_stride = group_stride
    p.k_tok_stride, p.v_tok_stride = tok_stride
    p.k_head_stride, p.v_head_stride = head_stride
    p.dtype = _dtype_code(q)
    p.B, p.nq, p.Hq, p.Hkv, p.D = B, nq, Hq, Hkv, D
    p.sb, p.kv_len, p.max_q_len = sb, kv_len, max_q_len
    p.causal = 1 if causal else 0
    p.lse_layout = lse_layout
    p.num_splits = num_splits
    p.softmax_scale = _scale_var.get()
    ws_bytes = lib.hyd_prefix_workspace_bytes(C.byref(p))
    ws = None
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws_bytes
    _lib.check(lib.hyd_prefix_attn_fwd(C.byref(p), _stream()))
    return out, lse


def flash_attention(q: Tensor, k: Tensor, v: Tensor, causal: bool = False) -> tuple[Tensor, Tensor]:
    """
    q: [b, seq q, qheads, dim]
    k: [b, seq k, kheads, dim]
    v: [b, seq k, kheads, dim]

    Returns (out [b, seq q, qheads, dim], softmax_lse [b, qheads, seq q] fp32), like
    flash.py:284-306.  Softmax scale dim**-0.5; causal masks are bottom-right aligned.
    """
    _require_gpu(q, k, v)
    assert q.ndim == 4 and k.ndim == 4 and v.ndim == 4, f"{q.shape} {k.shape} {v.shape}"
    assert k.shape == v.shape, f"{k.shape} {v.shape}"
    b, sq, hq, d = q.shape
    assert k.shape[0] == b and k.shape[3] == d, f"{q.shape} {k.shape}"
    dp = padded_head_dim(d)
    if dp != d:
        with true_head_dim_scale(d):
            out, lse = flash_attention(pad_head_dim(q, dp), pad_head_dim(k, dp), pad_head_dim(v, dp), causal)
        return out[..., :d].contiguous(), lse
    q = _q_contig(q)
    k, v = _lastdim_contig(k), _lastdim_contig(v)
    return prefix_attention(
        q, k, v, sb=b, kv_len=k.shape[1], group_stride=(k.stride(0), v.stride(0)),
        tok_stride=(k.stride(1), v.stride(1)), head_stride=(k.stride(2), v.stride(2)),
        B=b, nq=sq, causal=causal, lse_layout=HYD_LSE_BHQ, lse_shape=(b, hq, sq),
    )


def flash_attention_varlen(
    q: Tensor, k: Tensor, v: Tensor, cu_seqlens_q: Tensor, cu_seqlens_k: Tensor,
    max_seqlen_q: int, max_seqlen_k: int, causal: bool = False,
) -> tuple[Tensor, Tensor]:
    """
    q: [b*seq q, qheads, dim]
    k: [b*seq k, kheads, dim]
    v: [b*seq k, kheads, dim]

    Returns (out [total q, qheads, dim], softmax_lse [nseq, qheads, max_seqlen_q]) like
    flash.py:309-351 (flash-attn 2.3.6 layout; entries past a sequence's length are 0).
    """
    _require_gpu(q, k, v, cu_seqlens_q, cu_seqlens_k)
    assert q.ndim == 3 and k.ndim == 3 and v.ndim == 3
    assert k.shape == v.shape
    assert cu_seqlens_q.dtype == torch.int32 and cu_seqlens_k.dtype == torch.int32
    assert cu_seqlens_q.shape == cu_seqlens_k.shape
    nseq = cu_seqlens_q.shape[0] - 1
    tq, hq, d = q.shape
    dp = padded_head_dim(d)
    if dp != d:
        with true_head_dim_scale(d):
            out, lse = flash_attention_varlen(pad_head_dim(q, dp), pad_head_dim(k, dp), pad_head_dim(v, dp), cu_seqlens_q,
                                              cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal)
        return out[..., :d].contiguous(), lse
    q = _q_contig(q)
    k, v = _lastdim_contig(k), _lastdim_contig(v)
    return prefix_attention(
        q, k, v, sb=nseq, kv_len=int(max_seqlen_k), group_stride=(0, 0),
        tok_stride=(k.stride(0), v.stride(0)), head_stride=(k.stride(1), v.stride(1)),
        B=tq, nq=1, causal=causal, lse_layout=HYD_LSE_BHQ, lse_shape=(nseq, hq, int(max_seqlen_q)),
        cu_seqlens_k=cu_seqlens_k.contiguous(), cu_seqlens_q=cu_seqlens_q.contiguous(),
        max_q_len=int(max_seqlen_q),
    )


def fill_suffix_params(p: SuffixParams, q: Tensor, k: Tensor, v: Tensor, seq_len: Tensor | None, out: Tensor):
    b, nq, hq, d = q.shape
    p.q, p.out = q.data_ptr(), out.data_ptr()
    p.k = k.data_ptr() if k.numel() else None
    p.v = v.data_ptr() if v.numel() else None
    p.k_batch_stride, p.k_tok_stride, p.k_head_stride = k.stride(0), k.stride(1), k.stride(2)
    p.v_batch_stride, p.v_tok_stride, p.v_head_stride = v.stride(0), v.stride(1), v.stride(2)
    p.dtype = _dtype_code(q)
    p.B, p.nq, p.Hq, p.Hkv, p.D = b, nq, hq, k.shape[2], d
    p.kv_len = k.shape[1]
    p.softmax_scale = _scale_var.get()
    order = _order_var.get()
    if order is not None:
        if order.numel() != b or order.device != q.device:
            raise ValueError(f"seq_order has {order.numel()} entries on {order.device} for a batch of {b} on {q.device}")
        p.seq_order = order.data_ptr()
    keep = None
    if seq_len is not None:
        assert seq_len.shape == (b,), f"{seq_len.shape}"
        if seq_len.dtype == torch.int32:
            keep = seq_len.contiguous()
            p.seq_lens_i32 = keep.data_ptr()
        elif seq_len.dtype == torch.int64:
            keep = seq_len.contiguous()
            p.seq_lens_i64 = keep.data_ptr()
        else:
            keep = seq_len.to(torch.int32)
            p.seq_lens_i32 = keep.data_ptr()
    return keep


def flash_attention_seqlen(raw_q: Tensor, raw_k: Tensor, raw_v: Tensor, seq_len=None):
    """
    q shape: [batch, qseq_len, qheads, dim]
    k shape: [batch, kseq_len, kheads, dim]
    v shape: [batch, kseq_len, kheads, dim]

    Non-causal attention of every query over the first seq_len[b] keys of sequence b
    (flash.py:163-281).  Returns (out [b, q, h, d], lse [b, q, h] fp32, natural log).
    seq_len may be int32 or int64 (no cast kernel, cf. flash.py:220); None = all keys.
    """
    _require_gpu(raw_q, raw_k, raw_v, seq_len)
    assert raw_q.ndim == 4 and raw_k.ndim == 4 and raw_v.ndim == 4
    assert raw_k.shape == raw_v.shape
    assert raw_q.shape[-1] == raw_k.shape[-1], (
        f"Keys have head dim {raw_k.shape[-1]} but queries have head dim {raw_q.shape[-1]}"
    )
    d = raw_q.shape[-1]
    dp = padded_head_dim(d)
    if dp != d:
        with true_head_dim_scale(d):
            out, lse = flash_attention_seqlen(pad_head_dim(raw_q, dp), pad_head_dim(raw_k, dp), pad_head_dim(raw_v, dp), seq_len)
        return out[..., :d].contiguous(), lse
    lib = _lib.load()
    q = _q_contig(raw_q)
    k, v = _lastdim_contig(raw_k), _lastdim_contig(raw_v)
    out = torch.empty_like(q)
    lse = torch.empty(q.shape[:3], dtype=torch.float32, device=q.device)
    p = SuffixParams()
    keep = fill_suffix_params(p, q, k, v, seq_len, out)
    p.lse = lse.data_ptr()
    p.n_partials = 0
    _lib.check(lib.hyd_suffix_attn_fwd(C.byref(p), _stream()))
    del keep
    return out, lse
