"""
Shared, seed-driven input construction for parity tests and golden fixtures.

The hierarchy specs are the ones the reference's own operator test uses
(/root/reference/tests/test_attention.py:26-32: last entry = unique lengths,
earlier entries = shared levels), plus the literal BASELINE.json config 1 and a
few extra edge cases.  Inputs are generated with numpy's PCG64 (stable across
numpy versions) so that fixtures only need to store the seed and the expected
outputs.
"""

from __future__ import annotations

import numpy as np

# tests/test_attention.py:26-32
REFERENCE_SIZE_SPECS = [
    [[1], [10]],
    [[3], [6, 6]],
    [[3], [6, 7]],
    [[7, 7], [9, 10, 11, 4], [129, 2, 3, 4, 5, 6, 7, 128]],
    [[16384], [1, 128, 256]],
]
# tests/test_attention.py:17-19
REFERENCE_QHEADS = 8
REFERENCE_KVHEADS = [1, 8]
REFERENCE_DIM = 128


def _round(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round float32 values to the 16-bit storage dtype, return float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dtype == "f16":
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    u = (u + 0x7FFF + lsb) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def make_case(sizes, qheads: int, kvheads: int, dim: int, dtype: str, seed: int, nq: int = 1,
              force_seq_lens: bool = False):
    """Build the inputs of one hierarchy case exactly the way the reference test
    does (tests/test_attention.py:51-112) but from a numpy generator.

    Returns a dict of float32 numpy arrays (already rounded to `dtype`) and ints:
      q [B,nq,Hq,D], k/v [B,maxlen,Hkv,D], seq_lens (int32 [B] or None),
      shared_ks/shared_vs (4-D if equal lengths else packed 3-D),
      shared_cu_seq_lens (int32 or None), shared_max_seq_lens, use_varlens.
    """
    rng = np.random.default_rng(seed)

    def randn(*shape):
        return _round(rng.standard_normal(shape, dtype=np.float32), dtype)

    final = sizes[-1]
    B = len(final)
    q = randn(B, nq, qheads, dim)
    shared_ks, shared_vs, culens, maxlens, use_varlens = [], [], [], [], []
    for lens in sizes[:-1]:
        use_varlen = len(set(lens)) > 1
        use_varlens.append(use_varlen)
        if use_varlen:
            total = sum(lens)
            sk = randn(total, kvheads, dim)
            sv = randn(total, kvheads, dim)
            culens.append(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32))
            maxlens.append(int(max(lens)))
        else:
            sk = randn(len(lens), lens[0], kvheads, dim)
            sv = randn(len(lens), lens[0], kvheads, dim)
            culens.append(None)
            maxlens.append(None)
        shared_ks.append(sk)
        shared_vs.append(sv)
    maxlen = max(final)
    k = randn(B, maxlen, kvheads, dim)
    v = randn(B, maxlen, kvheads, dim)
    if len(set(final)) > 1 or force_seq_lens:
        seq_lens = np.asarray(final, dtype=np.int32)
    else:
        seq_lens = None
    return dict(
        q=q, k=k, v=v, seq_lens=seq_lens,
        shared_ks=shared_ks, shared_vs=shared_vs,
        shared_cu_seq_lens=culens, shared_max_seq_lens=maxlens, use_varlens=use_varlens,
        sizes=sizes, qheads=qheads, kvheads=kvheads, dim=dim, dtype=dtype, seed=seed, nq=nq,
    )


def golden_case_list():
    """(name, kwargs) for every committed golden fixture."""
    cases = []
    for si, sizes in enumerate(REFERENCE_SIZE_SPECS):
        for kvh in REFERENCE_KVHEADS:
            cases.append((f"ref_spec{si}_kv{kvh}_f16",
                          dict(sizes=sizes, qheads=REFERENCE_QHEADS, kvheads=kvh,
                               dim=REFERENCE_DIM, dtype="f16", seed=1000 + 10 * si + kvh)))
    # BASELINE.json config 1 literal: batch 4, prefix 64, suffix 8, 4 heads, dim 64
    cases.append(("c1_literal_f16", dict(sizes=[[64], [8, 8, 8, 8]], qheads=4, kvheads=4, dim=64,
                                         dtype="f16", seed=7, force_seq_lens=True)))
    cases.append(("c1_literal_bf16", dict(sizes=[[64], [8, 8, 8, 8]], qheads=4, kvheads=4, dim=64,
                                          dtype="bf16", seed=7, force_seq_lens=True)))
    # ragged unique lengths incl. length 1 and length == full cache; GQA g in {1,4,8}
    for g, kvh in [(1, 8), (4, 2), (8, 1)]:
        cases.append((f"ragged_g{g}_f16", dict(sizes=[[33], [5, 16, 3, 1]], qheads=8, kvheads=kvh,
                                               dim=128, dtype="f16", seed=50 + g)))
        cases.append((f"ragged_g{g}_bf16", dict(sizes=[[33], [5, 16, 3, 1]], qheads=8, kvheads=kvh,
                                                dim=128, dtype="bf16", seed=50 + g)))
    # two-level hierarchy shaped like BASELINE config 4, shrunk: 1x48 + 4x24, 8 completions
    cases.append(("two_level_f16", dict(sizes=[[48], [24] * 4, [7] * 8], qheads=8, kvheads=2, dim=128,
                                        dtype="f16", seed=91, force_seq_lens=True)))
    cases.append(("two_level_bf16", dict(sizes=[[48], [24] * 4, [7] * 8], qheads=8, kvheads=2, dim=128,
                                         dtype="bf16", seed=91, force_seq_lens=True)))
    return cases
