"""Python face of `hyd_rope_append_decode` (include/hydragen_hip.h): the fused RoPE + unique-KV append +
seq_lens kernel of the decode step (replaces /root/reference/hydragen/llama.py:236-262,485-501,565-569)."""

from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor

from . import _lib
from ._lib import RopeParams
from .flash import _dtype_code, _require_gpu, _stream


def rope_append_decode(q: Tensor, k: Tensor, v: Tensor, cos: Tensor, sin: Tensor, position_ids: Tensor,
                       shared_len: Tensor | None, k_cache: Tensor, v_cache: Tensor):
    """q [B,1,Hq,D], k/v [B,1,Hkv,D] (this step's projections), cos/sin fp32 [max_pos, D],
    position_ids int64 [B,1] absolute, shared_len int64 [B] or None, caches [maxB, maxS, Hkv, D].
    Returns (rotated q [B,1,Hq,D], seq_lens int32 [B]); k (rotated) and v are written into the caches
    at index position - shared_len."""
    _require_gpu(q, k, v, cos, sin, position_ids, k_cache, v_cache)
    lib = _lib.load()
    B, one, Hq, D = q.shape
    assert one == 1 and k.shape[:2] == (B, 1) and v.shape == k.shape
    assert q.stride(3) == 1 and q.stride(2) == D and k.stride(3) == 1 and k.stride(2) == D and v.stride(2) == D
    assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.stride(1) == 1
    assert position_ids.dtype == torch.int64 and position_ids.shape[0] == B
    assert B <= k_cache.shape[0]
    Hkv = k.shape[2]
    q_out = torch.empty((B, 1, Hq, D), dtype=q.dtype, device=q.device)
    seq_lens = torch.empty((B,), dtype=torch.int32, device=q.device)
    p = RopeParams()
    p.q, p.k, p.v, p.q_out = q.data_ptr(), k.data_ptr(), v.data_ptr(), q_out.data_ptr()
    p.k_cache, p.v_cache = k_cache.data_ptr(), v_cache.data_ptr()
    p.cos, p.sin = cos.data_ptr(), sin.data_ptr()
    p.position_ids = position_ids.data_ptr()
    if shared_len is not None:
        assert shared_len.dtype == torch.int64 and shared_len.shape == (B,)
        shared_len = shared_len.contiguous()
        p.shared_len = shared_len.data_ptr()
    p.seq_lens = seq_lens.data_ptr()
    p.q_batch_stride, p.k_batch_stride, p.v_batch_stride = q.stride(0), k.stride(0), v.stride(0)
    p.kc_batch_stride, p.kc_tok_stride, p.kc_head_stride = k_cache.stride(0), k_cache.stride(1), k_cache.stride(2)
    p.vc_batch_stride, p.vc_tok_stride, p.vc_head_stride = v_cache.stride(0), v_cache.stride(1), v_cache.stride(2)
    p.pos_stride, p.cs_stride = position_ids.stride(0), cos.stride(0)
    p.dtype, p.B, p.Hq, p.Hkv, p.D, p.cache_len = _dtype_code(q), B, Hq, Hkv, D, k_cache.shape[1]
    p.max_pos = cos.shape[0]
    _lib.check(lib.hyd_rope_append_decode(C.byref(p), _stream()))
    return q_out, seq_lens
