"""
CPU ORACLE (test infrastructure, NOT product code) -- numpy float64 restatement of
the reference's decomposed shared-prefix attention.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product path (`hydragen_amd/`) never does: it fails
loudly when the HIP library is missing.

Parity status: PINNED against the reference's own Python run in the authoring
container (`oracle/make_golden.py` imports `/root/reference/hydragen/attention.py`
and `flash.py`; the reference's Triton kernels run under TRITON_INTERPRET=1; the
un-vendored third-party `flash-attn v2.3.6` (requirements.txt:7) boundary is
replaced there by exact softmax attention because the reference holds no
known-answer vectors for it -- see DESIGN.md "Oracle").  The resulting vectors
live in `tests/golden/` and `tests/test_oracle.py` checks this file against them.

All functions take numpy arrays that already hold the *rounded* fp16/bf16 input
values (as float32/float64) and compute in float64, so the oracle is the
mathematical answer on identical inputs.

Reference lines followed (relative to /root/reference):
  combine_lse            hydragen/attention.py:21-43
  hydragen_attention     hydragen/attention.py:177-354
  flash_attention        hydragen/flash.py:284-306  (flash-attn semantics)
  flash_attention_varlen hydragen/flash.py:309-351
  flash_attention_seqlen hydragen/flash.py:163-281, hydragen/xformers_stuff.py:267-428
  pure-PyTorch listing   README.md:377-461
"""

from __future__ import annotations

import numpy as np

NEG_INF = -np.inf


def _f64(x):
    return np.asarray(x, dtype=np.float64)


def attention_lse(q, k, v, causal: bool = False, kv_lens=None, round_p=None):
    """Exact softmax attention returning (out, lse).

    round_p (callable or None): the reference's kernels round the un-normalised probabilities to the q dtype
    before P.V while the row sum keeps the un-rounded ones (xformers_stuff.py:388-391; flash-attn v2.3.6 does
    the same in its softmax -> P.V step).  Pass round_bf16 / round_fp16 to model that; None = exact.

    q [b, sq, hq, d]; k, v [b, sk, hkv, d]; GQA: q-head h reads kv-head h // (hq//hkv).
    scale = d**-0.5 (flash.py:293); lse = natural-log logsumexp of the scaled
    scores, layout [b, hq, sq] like flash-attn's softmax_lse (flash.py:295-306).
    causal=True uses flash-attn >= 2.1 bottom-right alignment: query i sees keys
    j <= i + (sk - sq)  (SURVEY K2c).
    kv_lens: optional per-batch number of valid keys (flash.py:220, xformers_stuff.py:274-279).
    """
    q, k, v = _f64(q), _f64(k), _f64(v)
    b, sq, hq, d = q.shape
    _, sk, hkv, _ = k.shape
    assert hq % hkv == 0
    g = hq // hkv
    scale = float(d) ** -0.5
    out = np.zeros((b, sq, hq, d), dtype=np.float64)
    lse = np.full((b, hq, sq), NEG_INF, dtype=np.float64)
    for bi in range(b):
        n = sk if kv_lens is None else int(kv_lens[bi])
        if n == 0:
            continue
        for h in range(hq):
            kk = k[bi, :n, h // g]  # [n, d]
            vv = v[bi, :n, h // g]
            s = (q[bi, :, h] @ kk.T) * scale  # [sq, n]
            if causal:
                i = np.arange(sq)[:, None]
                j = np.arange(n)[None, :]
                s = np.where(j <= i + (n - sq), s, NEG_INF)
            m = s.max(axis=1, keepdims=True)
            m = np.where(np.isfinite(m), m, 0.0)
            p = np.exp(s - m)
            l = p.sum(axis=1, keepdims=True)
            pr = p if round_p is None else round_p(p)
            with np.errstate(divide="ignore", invalid="ignore"):
                o = np.where(l > 0, (pr @ vv) / l, 0.0)
                ls = np.where(l[:, 0] > 0, m[:, 0] + np.log(l[:, 0]), NEG_INF)
            out[bi, :, h] = o
            lse[bi, h] = ls
    return out, lse


def combine_lse(outs, lses):
    """attention.py:21-43 -- out = sum_i out_i*exp(lse_i - max) / sum_i exp(lse_i - max).

    outs: list of [b, s, h, d]; lses: list of [b, s, h].
    """
    outs = np.stack([_f64(o) for o in outs])
    lses = np.stack([_f64(l) for l in lses])
    max_lse = lses.max(0)
    safe_max = np.where(np.isfinite(max_lse), max_lse, 0.0)
    adj = np.exp(lses - safe_max[None])
    den = adj.sum(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        agg = (outs * adj[..., None]).sum(0) / den[..., None]
    return agg


def flash_attention(q, k, v, causal: bool = False):
    """flash.py:284-306 -> (out [b,sq,hq,d], lse [b,hq,sq])."""
    return attention_lse(q, k, v, causal=causal)


def flash_attention_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal=False, round_p=None):
    """flash.py:309-351 -- packed q [sum_q, hq, d], k/v [sum_k, hkv, d].

    Returns out [sum_q, hq, d] and lse [nseq, hq, max_seqlen_q] (flash-attn 2.3.6
    layout, padded with 0 beyond each sequence's length; only the valid part is
    ever consumed: attention.py:333-338 uses uniform query counts).
    """
    q, k, v = _f64(q), _f64(k), _f64(v)
    nseq = len(cu_seqlens_q) - 1
    hq = q.shape[1]
    out = np.zeros_like(q)
    lse = np.zeros((nseq, hq, max_seqlen_q), dtype=np.float64)
    for i in range(nseq):
        q0, q1 = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
        k0, k1 = int(cu_seqlens_k[i]), int(cu_seqlens_k[i + 1])
        o, l = attention_lse(q[None, q0:q1], k[None, k0:k1], v[None, k0:k1], causal=causal, round_p=round_p)
        out[q0:q1] = o[0]
        lse[i, :, : q1 - q0] = l[0]
    return out, lse


def flash_attention_seqlen(q, k, v, seq_len, round_p=None):
    """flash.py:163-281 -- non-causal attention of every query row over the first
    seq_len[b] keys of sequence b.  Returns (out [b,q,h,d], lse [b,q,h]) with the
    natural-log LSE that flash.py:159-160 writes.
    """
    out, lse = attention_lse(q, k, v, causal=False, kv_lens=seq_len, round_p=round_p)
    return out, np.transpose(lse, (0, 2, 1))


def hydragen_attention(
    q,
    k,
    v,
    shared_ks,
    shared_vs,
    shared_cu_seq_lens,
    shared_max_seq_lens,
    use_varlens,
    seq_lens=None,
    round_partials=None,
    round_p=None,
):
    """attention.py:177-354.

    `round_p`: see attention_lse.  round_partials = round_p = round_bf16 and a final round_bf16 of the result is
    the REFERENCE'S OWN ARITHMETIC in float64 ("reference rounding model", reference_rounding_model below).

    Sequence i of the batch belongs to shared sequence i // (B / sb) at every
    level (attention.py:264-268).  `round_partials` (callable or None) is applied
    to each partial `out` before the merge; the reference rounds partials to the
    q dtype there (README.md:488-490) -- pass e.g. a bf16 rounding function to
    model that, or None for the exact answer.
    """
    q, k, v = _f64(q), _f64(k), _f64(v)
    b, nq, hq, d = q.shape
    rp = round_partials if round_partials is not None else (lambda x: x)
    outs, lses = [], []
    for sk, sv, scu, smax, use_varlen in zip(
        shared_ks, shared_vs, shared_cu_seq_lens, shared_max_seq_lens, use_varlens
    ):
        sk, sv = _f64(sk), _f64(sv)
        if not use_varlen:
            ns = sk.shape[0]
            assert b % ns == 0
            batched_q = q.reshape(ns, (b // ns) * nq, hq, d)  # attention.py:264-268
            so, sl = attention_lse(batched_q, sk, sv, round_p=round_p)
            so = so.reshape(b, nq, hq, d)
            if k.shape[1] == 0 and len(shared_ks) == 1:  # attention.py:273-274
                return so
            # "ns h (sps nq) -> (ns sps) nq h"   attention.py:276-280
            sl = sl.reshape(ns, hq, b // ns, nq).transpose(0, 2, 3, 1).reshape(b, nq, hq)
        else:
            ns = len(scu) - 1
            assert b % ns == 0
            qps = (b // ns) * nq
            cu_q = np.arange(ns + 1) * qps  # attention.py:295-311
            so, sl = flash_attention_varlen(q.reshape(b * nq, hq, d), sk, sv, cu_q, scu, qps, smax, round_p=round_p)
            so = so.reshape(b, nq, hq, d)
            if k.shape[1] == 0 and len(shared_ks) == 1:
                return so
            sl = sl.reshape(ns, hq, b // ns, nq).transpose(0, 2, 3, 1).reshape(b, nq, hq)
        outs.append(rp(so))
        lses.append(sl)

    if seq_lens is None:
        uo, ul = attention_lse(q, k, v, causal=True, round_p=round_p)  # attention.py:344-345
        ul = np.transpose(ul, (0, 2, 1))
    else:
        uo, ul = flash_attention_seqlen(q, k, v, seq_lens, round_p=round_p)  # attention.py:347
    outs.append(rp(uo))
    lses.append(ul)
    return combine_lse(outs, lses)


def hydragen_attention_nopad(q, k, v, shared_ks, shared_vs, seq_len=None, round_partials=None, round_p=None):
    """attention.py:357-392."""
    n = len(shared_ks)
    return hydragen_attention(
        q, k, v, shared_ks, shared_vs, [None] * n, [None] * n, [False] * n, seq_len, round_partials, round_p
    )


def reference_rounding_model(dtype: str, q, k, v, shared_ks, shared_vs, shared_cu_seq_lens, shared_max_seq_lens, use_varlens,
                             seq_lens=None):
    """The reference's OWN roundings, everything else float64: probabilities rounded to the q dtype before P.V
    (xformers_stuff.py:391; flash-attn), every partial `out` rounded to the q dtype on its way through HBM
    (attention.py:272, flash.py:254 -- the precision loss README.md:488-490 acknowledges), the merged result
    rounded to the q dtype (attention.py:120,147).  Its distance from the exact answer is the error the reference
    itself makes on these inputs: the yardstick of tests/test_reference_error_budget_gpu.py."""
    rnd = round_bf16 if dtype == "bf16" else round_fp16
    return rnd(hydragen_attention(q, k, v, shared_ks, shared_vs, shared_cu_seq_lens, shared_max_seq_lens, use_varlens,
                                  seq_lens, round_partials=rnd, round_p=rnd))


def nosharing_attention(q, k, v, shared_ks, shared_vs, shared_cu_seq_lens, use_varlens, seq_lens=None):
    """The undecomposed answer the reference's own test compares against
    (tests/test_attention.py:132-178): for every sequence, concatenate its shared
    slices and its unique slice and run plain non-causal attention.
    Requires nq == 1 (as in the reference test) unless seq_lens is None and the
    caller wants bottom-right-causal semantics over the unique part.
    """
    q, k, v = _f64(q), _f64(k), _f64(v)
    b, nq, hq, d = q.shape
    res = np.zeros((b, nq, hq, d))
    for i in range(b):
        ks, vs = [], []
        for sk, sv, scu, uv in zip(shared_ks, shared_vs, shared_cu_seq_lens, use_varlens):
            sk, sv = _f64(sk), _f64(sv)
            if uv:
                ns = len(scu) - 1
                si = i // (b // ns)
                ks.append(sk[int(scu[si]) : int(scu[si + 1])])
                vs.append(sv[int(scu[si]) : int(scu[si + 1])])
            else:
                ns = sk.shape[0]
                si = i // (b // ns)
                ks.append(sk[si])
                vs.append(sv[si])
        n = k.shape[1] if seq_lens is None else int(seq_lens[i])
        nshared = sum(x.shape[0] for x in ks)
        ks.append(k[i, :n])
        vs.append(v[i, :n])
        kk = np.concatenate(ks, 0)[None]
        vv = np.concatenate(vs, 0)[None]
        if seq_lens is None and nq > 1:
            # causal over the unique part only, all shared keys visible
            o, _ = attention_lse(q[i : i + 1], kk, vv, causal=True)
        else:
            o, _ = attention_lse(q[i : i + 1], kk, vv, causal=False)
        res[i] = o[0]
    return res


def rdiff(a, b, eps=1e-8):
    """hydragen/utils.py:13-15."""
    a, b = _f64(a), _f64(b)
    return 2 * np.abs(a - b) / (np.abs(a) + np.abs(b) + eps)


# ---------------------------------------------------------------------------
# dtype rounding helpers (inputs are rounded once, then everything is float64)
# ---------------------------------------------------------------------------

def round_fp16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float64)


def round_bf16(x):
    """Round-to-nearest-even float32 -> bfloat16 -> float64 (numpy has no bf16)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    u = (u + 0x7FFF + lsb) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def to_bits(x, dtype: str):
    """float array -> uint16 raw bits in the given 16-bit dtype ('f16'|'bf16')."""
    if dtype == "f16":
        return np.asarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    return (((u + 0x7FFF + lsb) >> 16) & 0xFFFF).astype(np.uint16)


def from_bits(u, dtype: str):
    u = np.asarray(u, dtype=np.uint16)
    if dtype == "f16":
        return u.view(np.float16).astype(np.float64)
    return (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
