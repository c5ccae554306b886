"""-m gpu: the token-row suffix kernel (csrc/suffix_attn.hip, `suffix_attn_rows_kernel`: the dominant kernel at BASELINE config 2)
against the float64 oracle on the shapes that SELECT it -- one query row per kv head (nq = 1, Hq = Hkv), kv heads a multiple of
the 64 / (D / 8) heads one wave instruction covers, at least 2048 (sequence, kv head) units, caches of at most 1024 rows -- which the
small cases of test_primitives_gpu.py never reach.  Every geometry of its launcher (1, 2 or 4 waves per sequence inside a workgroup,
a second workgroup row of heads, 2 or 4 waves sharing one sequence's tokens when a wave covers all of a token's heads), lengths around the 8-token chunk (0, 1, 7, 8, 9, 15, 16, 17, the cache's capacity), padding behind
a sequence's length poisoned with NaN / Inf (the kernel's requests are clamped, never predicated), 0 / 1 / 2 prefetched partials
and an fp32 slice behind them, the K|V arena layout the model allocates ([batch, K|V, rows, heads, dim]: strided views), int64
lengths.  Replaces flash_attention_seqlen + combine_lse (`/root/reference/hydragen/flash.py:163-281`, `attention.py:21-43`)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import _round
from tests.gpu_util import TORCH_DT, assert_close, assert_close_l2, dev

pytestmark = pytest.mark.gpu

EDGE_LENS = [0, 1, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33]


def _lens(rng, B, cap):
    sl = rng.integers(0, cap + 1, B).astype(np.int32)
    edge = [x for x in EDGE_LENS if x <= cap] + [cap, cap - 1]
    sl[:len(edge)] = edge
    sl[-1] = cap
    return sl


def _poison(x, sl, what):
    p = x.copy()
    for b in range(x.shape[0]):
        p[b, sl[b]:] = what if b % 2 else np.nan
    return p


# B x Hkv >= 2048 units each; waves per sequence = Hkv / (64 / (D / 8))
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("D,B,H,cap", [
    (128, 520, 4, 40),     # one wave covers a token's heads: 2 waves share a sequence (token split), 2 sequences per workgroup
    (128, 521, 4, 24),     # ... a cache too short to split (< 32 rows): 4 sequences per workgroup, the last workgroup a quarter full
    (128, 521, 4, 48),     # ... split, odd batch: the last workgroup's second pair of waves leaves before the barrier of the first
    (128, 2100, 4, 40),    # ... more than 2048 sequences: 4 waves share a sequence
    (128, 258, 8, 33),     # 2 waves per sequence: 2 sequences per workgroup
    (128, 129, 16, 64),    # 4 waves per sequence: one sequence per workgroup
    (128, 66, 32, 128),    # C2's geometry: two workgroups (rows of heads) per sequence
    (128, 45, 48, 20),     # 12 waves per sequence: three workgroup rows
    (64, 260, 8, 50),      # D = 64: 8 heads per wave instruction, 1 wave per sequence
    (64, 70, 32, 24),      # D = 64: 4 waves per sequence
    (256, 513, 4, 18),     # D = 256: 2 heads per wave instruction, 2 waves per sequence
    (256, 130, 16, 40),    # D = 256: 8 waves per sequence = two workgroup rows
])
def test_token_row_kernel_vs_oracle(dt, D, B, H, cap):
    from hydragen_amd.flash import flash_attention_seqlen

    rng = np.random.default_rng(D * 7 + B + H + cap)
    rnd = lambda *s: _round(rng.standard_normal(s, dtype=np.float32), dt)
    q, k, v = rnd(B, 1, H, D), rnd(B, cap, H, D), rnd(B, cap, H, D)
    sl = _lens(rng, B, cap)
    # the model's arena layout: a sequence's K rows, then its V rows (strided views), padding poisoned
    arena = torch.empty((B, 2, cap, H, D), dtype=TORCH_DT[dt], device="cuda:0")
    arena[:, 0] = dev(_poison(k, sl, np.inf), dt)
    arena[:, 1] = dev(_poison(v, sl, -np.inf), dt)
    out, lse = flash_attention_seqlen(dev(q, dt), arena[:, 0], arena[:, 1], seq_len=dev(sl))
    torch.cuda.synchronize()
    want, wlse = O.flash_attention_seqlen(q, k, v, sl)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all(), "padding leaked into the output"
    assert_close(got, want, dt, f"token-row kernel D={D} B={B} H={H}")
    fin = np.isfinite(wlse)
    gl = lse.cpu().numpy()
    assert np.abs(gl[fin] - wlse[fin]).max() < 2e-3 and np.all(np.isneginf(gl[~fin]))
    # empty sequences give exact zeros (attention.py:21-43: a partial over no keys drops out of the merge)
    assert not got[sl == 0].any()
    # int64 lengths take the same path (flash.py:220 casts; here no cast kernel)
    out64, _ = flash_attention_seqlen(dev(q, dt), arena[:, 0], arena[:, 1], seq_len=dev(sl.astype(np.int64)))
    assert torch.equal(out64, out)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,cap,parts", [
    (66, 32, 72, [("h", 1)]),                        # the decode step: one 16-bit prefix partial, prefetched under the stream
    (66, 32, 40, [("h", 1), ("h", 1)]),              # two-level hierarchy: both prefetched (NPRE = 2 instantiation)
    (129, 16, 24, [("h", 1), ("h", 1), ("f", 1)]),   # a third, fp32 partial behind the prefetched two
    (258, 8, 17, [("s", 3)]),                        # split-KV slices only: nothing prefetched
    (520, 4, 9, [("h", 1), ("s", 4)]),               # a TP = 8 shard: one prefetched partial + 4 fp32 slices
    (520, 4, 70, [("h", 1), ("s", 4)]),              # ... with caches long enough for the token split (2 waves per sequence)
    (2100, 4, 33, [("s", 2)]),                       # ... 4 waves per sequence
])
def test_token_row_kernel_folds_partials(dt, B, H, cap, parts):
    from hydragen_amd import _lib
    from hydragen_amd._lib import SuffixParams
    from hydragen_amd.flash import fill_suffix_params

    lib = _lib.load()
    D = 128
    rng = np.random.default_rng(B * 17 + H + cap + len(parts))
    rnd = lambda *s: _round(rng.standard_normal(s, dtype=np.float32), dt)
    q, k, v = rnd(B, 1, H, D), rnd(B, cap, H, D), rnd(B, cap, H, D)
    sl = _lens(rng, B, cap)
    tq, tsl = dev(q, dt), dev(sl)
    arena = torch.empty((B, 2, cap, H, D), dtype=TORCH_DT[dt], device="cuda:0")
    arena[:, 0] = dev(_poison(k, sl, np.inf), dt)
    arena[:, 1] = dev(_poison(v, sl, np.inf), dt)
    out = torch.empty_like(tq)
    sp = SuffixParams()
    keep = [fill_suffix_params(sp, tq, arena[:, 0], arena[:, 1], tsl, out)]
    rows = B * H
    al = lambda x: (x + 255) // 256 * 256
    outs, lses, n = [], [], 0
    for kind, cnt in parts:
        f32 = kind in ("f", "s")
        esz = 4 if f32 else 2
        ostride = al(rows * D * esz) if cnt > 1 else rows * D * esz
        lstride = al(rows * 4) if cnt > 1 else rows * 4
        ob = torch.zeros(cnt * ostride, dtype=torch.uint8, device=tq.device)
        lb = torch.zeros(cnt * lstride, dtype=torch.uint8, device=tq.device)
        for j in range(cnt):
            o = rng.standard_normal((B, 1, H, D), dtype=np.float32)
            o = o if f32 else _round(o, dt)
            l = (rng.standard_normal((B, 1, H)) * 2.0 + 3.0).astype(np.float32)
            l[rng.random((B, 1, H)) < 0.05] = -np.inf  # a partial over no keys drops out exactly
            o[~np.isfinite(l)] = 0.0
            t = torch.from_numpy(o).to(tq.device).to(torch.float32 if f32 else TORCH_DT[dt]).contiguous()
            ob[j * ostride:j * ostride + rows * D * esz] = t.view(torch.uint8).flatten()
            lb[j * lstride:j * lstride + rows * 4] = torch.from_numpy(l).to(tq.device).contiguous().view(torch.uint8).flatten()
            outs.append(o)
            lses.append(l)
        sp.partials[n].out, sp.partials[n].lse, sp.partials[n].count, sp.partials[n].is_f32 = ob.data_ptr(), lb.data_ptr(), cnt, int(f32)
        keep += [ob, lb]
        n += 1
    sp.n_partials = n
    _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    so, slse = O.flash_attention_seqlen(q, k, v, sl)
    so = np.where(np.isfinite(slse)[..., None], so, 0.0)
    want = O.combine_lse(outs + [so], lses + [slse])
    ok = np.isfinite(np.stack(lses + [slse]).max(0))
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all() and ok.any()
    assert_close_l2(got[ok], want[ok], dt, f"{parts} lens {sl[:8]}")
