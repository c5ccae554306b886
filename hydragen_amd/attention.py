"""
The Hydragen attention operator with the reference's signatures
(/root/reference/hydragen/attention.py): `hydragen_attention` (:177-354),
`hydragen_attention_nopad` (:357-392), `combine_lse` (:154-174).

Decode (the hot path) is ONE C call, `hyd_decode_attn_fused`: a prefix pass per shared level
on the matrix cores, then the suffix pass whose epilogue performs the log-sum-exp merge --
2 kernel launches for the usual single-prefix case, no LSE re-layout copies, no scratch
allocation inside the library.
"""

from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional

import torch
from torch import Tensor

from . import _lib
from . import flash as _flash
from ._lib import (HYD_F32, HYD_LSE_BQH, HYD_MAX_LEVELS, HYD_PHASE_ALL, HYD_PHASE_MERGE, HYD_PHASE_SHARED,
                   HYD_PHASE_UNIQUE_PARTIAL, DecodeParams)
from .flash import (
    _dtype_code, _lastdim_contig, _q_contig, _require_gpu, _stream, fill_suffix_params, prefix_attention,
)


def combine_lse(outs: list[Tensor], lses: list[Tensor], enable_triton: bool = True):
    """
    Merge attention results using log-sum-exp metadata (attention.py:154-174).

    Args:
        outs: Attention results, list of [batch, seq_len, qheads, hdim]
        lses: Log-sum-exps of outs, corresponding list of [batch, seq_len, qheads]
        enable_triton: accepted for signature compatibility; one HIP kernel handles any
            number of partials, any head dim and fp16/bf16/fp32 (attention.py:21-43 semantics).
    """
    assert len(outs) == len(lses) and len(outs) > 0
    _require_gpu(*outs, *lses)
    lib = _lib.load()
    ref = outs[0]
    for o, l in zip(outs, lses):
        assert o.shape == ref.shape, f"{o.shape} {ref.shape}"
        assert l.shape == ref.shape[:-1], f"{l.shape} {ref.shape}"
        assert o.dtype == ref.dtype
    if ref.dtype == torch.float32:
        dt = HYD_F32
    else:
        dt = _dtype_code(ref)
    outs_c = [o.contiguous() for o in outs]
    lses_c = [l.contiguous().float() for l in lses]
    n = len(outs_c)
    D = ref.shape[-1]
    rows = ref.numel() // D if D else 0
    res = torch.empty(ref.shape, dtype=ref.dtype, device=ref.device)
    op = (C.c_void_p * n)(*[o.data_ptr() for o in outs_c])
    lp = (C.c_void_p * n)(*[l.data_ptr() for l in lses_c])
    _lib.check(lib.hyd_combine_lse(op, lp, n, rows, D, dt, res.data_ptr(), None, _stream()))
    return res


def _fill_level(lv, sk: Tensor, sv: Tensor, scu, smax, use_varlen: bool, b: int):
    if not use_varlen:
        assert sk.ndim == 4, f"{sk.shape}"
        ns = sk.shape[0]
        assert b % ns == 0, f"{b} {ns}"
        lv.k, lv.v = sk.data_ptr(), sv.data_ptr()
        lv.cu_seqlens_k = None
        lv.k_group_stride, lv.k_tok_stride, lv.k_head_stride = sk.stride(0), sk.stride(1), sk.stride(2)
        lv.v_group_stride, lv.v_tok_stride, lv.v_head_stride = sv.stride(0), sv.stride(1), sv.stride(2)
        lv.sb, lv.kv_len = ns, sk.shape[1]
    else:
        assert sk.ndim == 3, f"{sk.shape}"
        assert scu is not None and scu.dtype == torch.int32
        ns = scu.shape[0] - 1
        assert b % ns == 0, f"{b} {ns}"
        lv.k, lv.v = sk.data_ptr(), sv.data_ptr()
        lv.cu_seqlens_k = scu.data_ptr()
        lv.k_group_stride = lv.v_group_stride = 0
        lv.k_tok_stride, lv.k_head_stride = sk.stride(0), sk.stride(1)
        lv.v_tok_stride, lv.v_head_stride = sv.stride(0), sv.stride(1)
        lv.sb, lv.kv_len = ns, int(smax)


def hydragen_attention(
    q: Tensor,
    k: Tensor,
    v: Tensor,
    shared_ks: list[Tensor],
    shared_vs: list[Tensor],
    shared_cu_seq_lens: list[Tensor | None],
    shared_max_seq_lens: list[int | None],
    use_varlens: list[bool],
    seq_lens: Tensor | None = None,
):
    """
    Computes Hydragen attention (attention decomposition + inter-sequence batching); same
    contract as attention.py:177-354.

    Args:
        q: attention queries, shape [batch, qlen, qheads, head_dim]
        k, v: unique-per-sequence keys/values, shape [batch, kvlen, kvheads, head_dim]
        shared_ks / shared_vs: per shared level, [sbatch, slen, kvheads, head_dim] when
            use_varlens[i] is False, else packed [total_slen, kvheads, head_dim].  sbatch must
            divide batch; sequence i uses shared sequence i // (batch / sbatch).
        shared_cu_seq_lens: per level int32 [sbatch+1] cumulative lengths (varlen levels) or None
        shared_max_seq_lens: per level max length (varlen levels) or None
        use_varlens: per level, whether the packed format is used
        seq_lens: lengths of the unique KVs (right padded).  If None the unique part is
            attended causally (bottom-right aligned), as the reference does via flash-attn.
    """
    assert q.ndim == 4, f"{q.shape}"
    assert k.ndim == 4, f"{k.shape}"
    assert v.ndim == 4, f"{v.shape}"
    assert k.shape == v.shape
    assert (
        len(shared_ks) == len(shared_vs) == len(shared_cu_seq_lens) == len(shared_max_seq_lens) == len(use_varlens)
    )
    for sk, sv in zip(shared_ks, shared_vs):
        assert sk.shape == sv.shape, f"{sk.shape} {sv.shape}"
    _require_gpu(q, k, v, *shared_ks, *shared_vs, seq_lens)
    n_levels = len(shared_ks)

    b, nq, hq, d = q.shape
    dp = _flash.padded_head_dim(d)
    if dp != d:  # zero-padded to the kernels' head dim, true scale (flash.py: "head dims other than ...")
        pad = lambda t: _flash.pad_head_dim(t, dp)
        with _flash.true_head_dim_scale(d):
            out = hydragen_attention(pad(q), pad(k), pad(v), [pad(x) for x in shared_ks], [pad(x) for x in shared_vs],
                                     shared_cu_seq_lens, shared_max_seq_lens, use_varlens, seq_lens)
        return out[..., :d].contiguous()
    q = _q_contig(q)
    k, v = _lastdim_contig(k), _lastdim_contig(v)
    shared_ks = [_lastdim_contig(x) for x in shared_ks]
    shared_vs = [_lastdim_contig(x) for x in shared_vs]

    # seq_lens None means "causal over the unique part" in the reference (attention.py:343-345);
    # with a single query the bottom-right-aligned causal mask hides nothing, which is the decode case.
    fused_ok = seq_lens is not None or nq == 1 or k.shape[1] == 0
    if fused_ok and n_levels <= HYD_MAX_LEVELS:
        return _decode_fused(q, k, v, shared_ks, shared_vs, shared_cu_seq_lens, shared_max_seq_lens,
                             use_varlens, seq_lens)

    # The general form, as the reference spells it for any number of levels (attention.py:250-352): one prefix pass per
    # level, one pass over the unique K/V, an N-way log-sum-exp merge.  Taken for the unique-suffix prefill (causal
    # MFMA pass over the unique K/V) and for hierarchies deeper than the one-call operator's HYD_MAX_LEVELS levels
    # (decode: the suffix kernel with its LSE).  No host synchronisation: capture-safe like the one-call form.
    outs, lses = [], []
    for sk, sv, scu, smax, use_varlen in zip(shared_ks, shared_vs, shared_cu_seq_lens, shared_max_seq_lens,
                                             use_varlens):
        o, l = _level_pass(q, sk, sv, scu, smax, use_varlen)
        outs.append(o)
        lses.append(l)
    kvh = k.shape[2]
    assert k.shape[0] == b and hq % kvh == 0
    if fused_ok:
        if k.shape[1] > 0:
            uo, ul = _flash.flash_attention_seqlen(q, k, v, seq_lens)
            outs.append(uo)
            lses.append(ul)
    else:
        uo, ul = prefix_attention(
            q, k, v, sb=b, kv_len=k.shape[1], group_stride=(k.stride(0), v.stride(0)),
            tok_stride=(k.stride(1), v.stride(1)), head_stride=(k.stride(2), v.stride(2)),
            B=b, nq=nq, causal=True, lse_layout=HYD_LSE_BQH, lse_shape=(b, nq, hq),
        )
        outs.append(uo)
        lses.append(ul)
    return _combine_many(outs, lses)


_COMBINE_MAX = 64  # partials one hyd_combine_lse launch merges (csrc/hyd_kernels.h kMaxCombine)


def _combine_many(outs: list[Tensor], lses: list[Tensor]) -> Tensor:
    """combine_lse for any number of partials: groups of at most _COMBINE_MAX are merged into (out, merged LSE) pairs
    until one group is left (the merge is associative: attention.py:21-43)."""
    while len(outs) > _COMBINE_MAX:
        nouts, nlses = [], []
        for i in range(0, len(outs), _COMBINE_MAX):
            go, gl = outs[i:i + _COMBINE_MAX], lses[i:i + _COMBINE_MAX]
            if len(go) == 1:
                nouts.append(go[0])
                nlses.append(gl[0])
                continue
            lib = _lib.load()
            ref = go[0]
            oc = [o.contiguous() for o in go]
            lc = [l.contiguous().float() for l in gl]
            res = torch.empty_like(oc[0])
            rl = torch.empty(ref.shape[:-1], dtype=torch.float32, device=ref.device)
            op = (C.c_void_p * len(oc))(*[o.data_ptr() for o in oc])
            lp = (C.c_void_p * len(lc))(*[l.data_ptr() for l in lc])
            dt = HYD_F32 if ref.dtype == torch.float32 else _dtype_code(ref)
            _lib.check(lib.hyd_combine_lse(op, lp, len(oc), ref.numel() // ref.shape[-1], ref.shape[-1], dt,
                                           res.data_ptr(), rl.data_ptr(), _stream()))
            nouts.append(res)
            nlses.append(rl)
        outs, lses = nouts, nlses
    return combine_lse(outs, lses)


def _level_pass(q, sk, sv, scu, smax, use_varlen):
    b, nq, hq, d = q.shape
    if not use_varlen:
        ns = sk.shape[0]
        assert b % ns == 0, f"{b} {ns}"
        return prefix_attention(
            q, sk, sv, sb=ns, kv_len=sk.shape[1], group_stride=(sk.stride(0), sv.stride(0)),
            tok_stride=(sk.stride(1), sv.stride(1)), head_stride=(sk.stride(2), sv.stride(2)),
            B=b, nq=nq, causal=False, lse_layout=HYD_LSE_BQH, lse_shape=(b, nq, hq),
        )
    ns = scu.shape[0] - 1
    assert b % ns == 0, f"{b} {ns}"
    return prefix_attention(
        q, sk, sv, sb=ns, kv_len=int(smax), group_stride=(0, 0),
        tok_stride=(sk.stride(0), sv.stride(0)), head_stride=(sk.stride(1), sv.stride(1)),
        B=b, nq=nq, causal=False, lse_layout=HYD_LSE_BQH, lse_shape=(b, nq, hq), cu_seqlens_k=scu,
    )


# Marshalled `hyd_decode_params` + workspace per (tensor set, shapes, stream, thread): an eager call then costs one dict
# lookup, one `torch.empty_like` and one C call instead of ~20 us of ctypes field stores (under HIP graphs the host path
# does not run at all).  An entry's struct is mutated on every hit (`suffix.out`) and its scratch workspace is written
# by the launch, so entries are private to the (stream, thread) that made them: two streams or two threads never share
# a struct or a workspace.  Values keep alive the scratch workspace and the (contiguous) length tensors the struct
# points at; a stale entry can at worst describe memory the caller no longer passes in -- it is then never looked up
# again.
_PARAM_CACHE: dict = {}
_PARAM_CACHE_MAX = 64
_PARAM_CACHE_MAX_BYTES = 256 << 20  # scratch held by cached entries
_param_cache_bytes = 0


# ---- two-stream form --------------------------------------------------------------------------------------------------
# The prefix passes (matrix-core bound, ~no HBM traffic) and the suffix pass (HBM bound, ~no matrix-core work) do not
# depend on each other; the reference issues them one after the other (attention.py:250-352).  Here the shared phase
# CAN run on a side stream, limited to TWO_STREAM_PREFIX_CUS persistent workgroups, beside the unique phase on the
# caller's stream, with a log-sum-exp combine behind the join (hyd_decode_params.phase, include/hydragen_hip.h).
# Measured on MI355X at C2 (profiles/r03_overlap_graph.md, profiles/r03_two_stream.md): the passes do overlap -- a graph
# replay timed ALONE drops from 220.6 to 200.5 us at suffix 64, 378 -> 361 at 128, 100.8 -> 90.3 at 16 -- but each of
# the two cross-queue edges (fork, join) costs ~10 us whenever the queues are busy (~25 us for the pair issued
# eagerly): replays that follow each other run at 220.7 vs 218.0 us per step, and the Llama-2-7B decode graph with the
# form inside every layer at 26.96 vs 26.38 ms per step.  So the default is "off": the form is an option for callers
# whose replays are isolated (the reference's own flush-between-iterations protocol), "auto" turns it on only while a
# HIP graph is being captured and only for shapes whose prefix work is worth hiding.
TWO_STREAM_PREFIX_CUS = 128
_two_stream_mode = "off"
_side_streams: dict = {}


def set_two_stream(mode: str) -> str:
    """"off" (default), "auto" (while capturing a HIP graph, for shapes with >= 4 GFLOP of prefix work) or "on"
    (whenever the shapes have both parts).  Returns the previous mode."""
    global _two_stream_mode
    assert mode in ("auto", "on", "off"), mode
    prev, _two_stream_mode = _two_stream_mode, mode
    return prev


_f32_partials = False


def set_f32_partials(on: bool) -> bool:
    """Keep every unsplit shared level's partial result in fp32 instead of the 16-bit dtype (hyd_decode_params.f32_partials):
    one rounding less before the merge (C2, bf16: relative L2 error 2.85e-3 -> see bench.py `accuracy`) for 2 more bytes
    per output element written and read back.  Default off: the reference itself merges 16-bit partials
    (README.md:488-490).  Returns the previous setting."""
    global _f32_partials
    prev, _f32_partials = _f32_partials, bool(on)
    return prev


def _side_stream(device) -> "torch.cuda.Stream":
    key = (device.index if device.index is not None else torch.cuda.current_device(), threading.get_ident())
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def _want_two_stream(q, k, shared_ks, shared_max_seq_lens, use_varlens, capturing: bool) -> bool:
    if _two_stream_mode == "off" or not shared_ks or k.shape[1] == 0:
        return False
    if _two_stream_mode == "on":
        return True
    if not capturing or k.shape[1] < 16:  # a unique cache that short cannot hide a prefix pass
        return False
    b, nq, hq, d = q.shape
    keys = sum(int(m) if uv else sk.shape[1] for sk, m, uv in zip(shared_ks, shared_max_seq_lens, use_varlens))
    return 4.0 * b * nq * hq * d * keys >= 4.0e9  # prefix flops worth hiding (C2: 34e9, C1: 1e5)


def _launch_decode(lib, p, two_stream: bool, stream: int):
    """Issue the operator described by `p`: one call, or the three calls of the two-stream form."""
    if not two_stream:
        p.phase, p.shared_max_workgroups = HYD_PHASE_ALL, 0
        p.single_launch_small = 1  # problems that are launch latency, not work, run as one kernel (hydragen_hip.h)
        _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))
        return
    p.single_launch_small = 0
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)
    p.shared_max_workgroups = TWO_STREAM_PREFIX_CUS
    side.wait_stream(main)
    try:
        p.phase = HYD_PHASE_SHARED
        _lib.check(lib.hyd_decode_attn_fused(C.byref(p), side.cuda_stream))
        p.phase = HYD_PHASE_UNIQUE_PARTIAL
        _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))
    finally:
        main.wait_stream(side)  # always join: a capture must not end with a dangling branch
    p.phase = HYD_PHASE_MERGE
    _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))


def _tensor_key(t):
    return None if t is None else (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)


def _decode_fused(q, k, v, shared_ks, shared_vs, shared_cu_seq_lens, shared_max_seq_lens, use_varlens, seq_lens):
    lib = _lib.load()
    b, nq, hq, d = q.shape
    out = torch.empty_like(q)
    # no cache while capturing (the scratch belongs to the graph's private pool) or with a head-dim override (padded temporaries)
    capturing = torch.cuda.is_current_stream_capturing()
    uncached = capturing or _flash.current_softmax_scale() != 0.0
    order = _flash.current_seq_order()  # schedule hint (flash.seq_order); part of the cached block's identity, kept alive with it
    two_stream = _want_two_stream(q, k, shared_ks, shared_max_seq_lens, use_varlens, capturing)
    key = None
    stream = _stream()
    if not uncached:
        key = (_tensor_key(q), _tensor_key(k), _tensor_key(v), _tensor_key(seq_lens), _tensor_key(order), q.device.index, stream, threading.get_ident(), _f32_partials,
               tuple(_tensor_key(x) for x in shared_ks), tuple(_tensor_key(x) for x in shared_vs),
               tuple(_tensor_key(x) for x in shared_cu_seq_lens), tuple(shared_max_seq_lens), tuple(use_varlens))
        hit = _PARAM_CACHE.get(key)
        if hit is not None:
            p = hit[0]
            p.suffix.out = out.data_ptr()
            _launch_decode(lib, p, two_stream, stream)
            return out
    p = DecodeParams()
    keep = [fill_suffix_params(p.suffix, q, k, v, seq_lens, out), order]
    p.n_levels = len(shared_ks)
    p.f32_partials = 1 if _f32_partials else 0
    for i, (sk, sv, scu, smax, uv) in enumerate(
        zip(shared_ks, shared_vs, shared_cu_seq_lens, shared_max_seq_lens, use_varlens)
    ):
        if uv:
            scu = scu.contiguous()
            keep.append(scu)
        _fill_level(p.levels[i], sk, sv, scu, smax, uv, b)
    if p.n_levels == 0:
        assert k.shape[1] > 0, "no shared levels and no unique keys"
    ws_bytes = lib.hyd_decode_workspace_bytes(C.byref(p))
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        keep.append(ws)
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws_bytes
    if two_stream and not lib.hyd_decode_two_stream_ok(C.byref(p)):
        two_stream = False
    _launch_decode(lib, p, two_stream, stream)
    # cache only when every pointer in `p` refers to caller-owned memory or to tensors `keep` holds on to
    cacheable = key is not None and (seq_lens is None or seq_lens.dtype in (torch.int32, torch.int64)) and \
        all(x is None or x.is_contiguous() for x in shared_cu_seq_lens) and (seq_lens is None or seq_lens.is_contiguous())
    if cacheable and ws_bytes <= _PARAM_CACHE_MAX_BYTES // 4:
        global _param_cache_bytes
        while _PARAM_CACHE and (len(_PARAM_CACHE) >= _PARAM_CACHE_MAX or _param_cache_bytes + ws_bytes > _PARAM_CACHE_MAX_BYTES):
            _param_cache_bytes -= _PARAM_CACHE.pop(next(iter(_PARAM_CACHE)))[2]
        _PARAM_CACHE[key] = (p, keep, ws_bytes)
        _param_cache_bytes += ws_bytes
    return out


def hydragen_attention_nopad(
    q: Tensor,
    k: Tensor,
    v: Tensor,
    shared_ks: List[Tensor],
    shared_vs: List[Tensor],
    seq_len: Optional[Tensor] = None,
):
    """
    Hydragen attention when no shared level needs padding (attention.py:357-392).

    Args:
        q: [batch, qlen, qheads, head_dim]
        k, v: unique-per-sequence keys/values [batch, kvlen, kvheads, head_dim]
        shared_ks / shared_vs: list of [sbatch, slen, kvheads, head_dim]; sbatch divides batch
        seq_len: lengths of the unique KVs (right padded); None = no padding
    """
    n = len(shared_ks)
    return hydragen_attention(
        q, k, v,
        shared_ks=shared_ks, shared_vs=shared_vs,
        shared_cu_seq_lens=[None] * n, shared_max_seq_lens=[None] * n, use_varlens=[False] * n,
        seq_lens=seq_len,
    )
