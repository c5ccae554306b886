"""
CPU ORACLE / BASELINE (test infrastructure, NOT product code): the reference algorithm restated
in plain torch CPU ops -- exactly the pure-PyTorch listing of /root/reference/README.md:377-461
(decomposed prefix + suffix + combine_lse) and its no-sharing counterpart
(tests/test_attention.py:132-178).  bench.py times it on the host cores as `cpu_baseline`
(kind "port": the reference itself has no CPU path and its Python does not travel).
Parity: checked against tests/golden by tests/test_oracle.py.
"""

from __future__ import annotations

import torch
from torch import Tensor


def attention(q: Tensor, k: Tensor, v: Tensor, kv_lens: Tensor | None = None):
    """README.md:381-391 'some fast attention primitive that also returns LSEs' in fp32 torch.
    q [b, sq, hq, d]; k, v [b, sk, hkv, d] -> out [b, sq, hq, d], lse [b, sq, hq].
    Many query rows per KV (the folded prefix pass) go through matmul; a handful of rows per sequence
    (the decode suffix pass: b*hkv tiny problems) go through broadcast multiply-reduce, which is what a
    CPU does well there (batched 1xD matmuls are overhead-bound)."""
    b, sq, hq, d = q.shape
    sk, hkv = k.shape[1], k.shape[2]
    g = hq // hkv
    qf = q.float().view(b, sq, hkv, g, d).permute(0, 2, 3, 1, 4).reshape(b, hkv, g * sq, d)
    kf = k.float().permute(0, 2, 1, 3)  # b hkv sk d
    vf = v.float().permute(0, 2, 1, 3)
    small = g * sq <= 8
    if small:
        s = (qf[:, :, :, None, :] * kf[:, :, None, :, :]).sum(-1) * (d ** -0.5)  # b hkv rows sk
    else:
        s = torch.matmul(qf, kf.transpose(-1, -2)) * (d ** -0.5)
    if kv_lens is not None:
        mask = torch.arange(sk)[None, :] >= kv_lens[:, None]  # [b, sk]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    if small:
        o = (p[..., None] * vf[:, :, None, :, :]).sum(-2)
    else:
        o = torch.matmul(p, vf)  # [b, hkv, g*sq, d]
    o = o.view(b, hkv, g, sq, d).permute(0, 3, 1, 2, 4).reshape(b, sq, hq, d)
    lse = lse.view(b, hkv, g, sq).permute(0, 3, 1, 2).reshape(b, sq, hq)
    return o, lse


def combine_lse(outs, lses):
    """hydragen/attention.py:21-43."""
    outs = torch.stack(outs)
    lses = torch.stack(lses)
    max_lse = lses.max(0).values
    adj = (lses - max_lse[None]).exp()
    den = adj.sum(0)
    return (outs * adj.unsqueeze(-1)).sum(0) / den.unsqueeze(-1)


def hydragen_attention_nopad(q, k, v, shared_ks, shared_vs, seq_len=None):
    """README.md:413-461 generalised to several uniform levels (hydragen/attention.py:250-352)."""
    b, nq, hq, d = q.shape
    outs, lses = [], []
    for sk, sv in zip(shared_ks, shared_vs):
        ns = sk.shape[0]
        bq = q.reshape(ns, (b // ns) * nq, hq, d)
        o, l = attention(bq, sk, sv)
        outs.append(o.reshape(b, nq, hq, d))
        lses.append(l.reshape(b, nq, hq))
    if k.shape[1] > 0:
        o, l = attention(q, k, v, kv_lens=seq_len)
        outs.append(o)
        lses.append(l)
    return combine_lse(outs, lses)


def nosharing_attention(q, k, v, shared_k, shared_v, seq_len=None):
    """No-sharing SDPA over concatenated KV; the prefix is expanded with stride 0."""
    b = q.shape[0]
    P = shared_k.shape[1]
    kk = torch.cat([shared_k.expand(b, -1, -1, -1), k], dim=1)
    vv = torch.cat([shared_v.expand(b, -1, -1, -1), v], dim=1)
    lens = None if seq_len is None else seq_len + P
    o, _ = attention(q, kk, vv, kv_lens=lens)
    return o
