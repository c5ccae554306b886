"""-m gpu: `hyd_suffix_attn_fwd` with prefix partials handed in directly (the fused combine of attention.py:352 /
attention.py:21-43 inside the suffix pass), on both suffix kernels.

The grouped-query kernel folds a wave's partials UNDER its K/V stream -- one per key step, through an asm-owned register
buffer -- and fetches what is left behind the loop; which partial takes which route depends on the number of partials,
the number of 32-key steps a wave has, the waves per unit and whether the caller also wants the suffix pass's own LSE
(then nothing may be folded before the keys end).  Every route against the float64 oracle: random normalised partials
(16-bit, fp32, stacked fp32 slices as a split prefix pass leaves them) + the oracle's suffix attention, merged by the
oracle's combine_lse."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import _round
from tests.gpu_util import TORCH_DT, assert_close_l2, dev

pytestmark = pytest.mark.gpu


def _al(x):
    return (x + 255) // 256 * 256


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("want_lse", [False, True])
@pytest.mark.parametrize("B,Hq,Hkv,mk,lens,parts", [
    # parts: list of (kind, count): "h" = one 16-bit partial, "f" = one fp32 partial, ("s", n) = n stacked fp32 slices
    (6, 8, 1, 40, [40, 1, 0, 33, 7, 32], [("h", 1)]),                        # few units: 4 waves per unit, one partial
    (6, 8, 1, 300, [300, 1, 0, 33, 129, 64], [("s", 5)]),                     # 5 slices over 4 waves: one wave folds two
    (5, 16, 2, 70, [70, 64, 3, 0, 32], [("h", 1), ("f", 1), ("s", 3), ("h", 1)]),  # mixed types, 6 partials
    (3, 32, 8, 20, [20, 1, 0], [("s", 16)]),                                  # C3-like: 16 slices, one key step per wave at most
    (300, 8, 2, 100, None, [("s", 2), ("h", 1)]),                            # many units: one wave per unit, 2 kv heads per workgroup
    (200, 64, 8, 150, None, [("s", 7)]),                                      # 8 kv heads per workgroup; 7 partials on 1..5 key steps
    (200, 4, 4, 90, None, [("s", 3), ("h", 1)]),                             # one row per unit: the dot-product kernel
])
def test_suffix_kernels_fold_partials_of_every_kind(dt, want_lse, B, Hq, Hkv, mk, lens, parts):
    from hydragen_amd import _lib
    from hydragen_amd._lib import SuffixParams
    from hydragen_amd.flash import fill_suffix_params

    lib = _lib.load()
    D = 128
    rng = np.random.default_rng(B * 131 + Hq + mk + len(parts))
    rnd = lambda *s: _round(rng.standard_normal(s, dtype=np.float32), dt)
    q, k, v = rnd(B, 1, Hq, D), rnd(B, mk, Hkv, D), rnd(B, mk, Hkv, D)
    sl = np.asarray(lens, dtype=np.int32) if lens is not None else rng.integers(0, mk + 1, B).astype(np.int32)
    if lens is None:
        sl[0], sl[-1] = mk, 1
    tq, tk, tv, tsl = dev(q, dt), dev(k, dt), dev(v, dt), dev(sl)
    out = torch.empty_like(tq)
    lse = torch.empty((B, 1, Hq), dtype=torch.float32, device=tq.device)
    sp = SuffixParams()
    keep = [fill_suffix_params(sp, tq, tk, tv, tsl, out)]
    if want_lse:
        sp.lse = lse.data_ptr()
    rows = B * Hq
    outs, lses = [], []  # what the oracle merges
    n = 0
    for kind, cnt in parts:
        f32 = kind in ("f", "s")
        esz = 4 if f32 else 2
        ostride = _al(rows * D * esz) if cnt > 1 else rows * D * esz
        lstride = _al(rows * 4) if cnt > 1 else rows * 4
        ob = torch.zeros(cnt * ostride, dtype=torch.uint8, device=tq.device)
        lb = torch.zeros(cnt * lstride, dtype=torch.uint8, device=tq.device)
        for j in range(cnt):
            o = rng.standard_normal((B, 1, Hq, D), dtype=np.float32)
            o = o if f32 else _round(o, dt)
            l = (rng.standard_normal((B, 1, Hq)) * 2.0 + 3.0).astype(np.float32)
            l[rng.random((B, 1, Hq)) < 0.05] = -np.inf  # a partial over no keys (an empty ragged group): drops out exactly
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
    ok = np.isfinite(np.stack(lses + [slse]).max(0))  # rows where at least one partial or one key exists
    got = out.float().cpu().numpy()
    assert ok.any()
    assert_close_l2(got[ok], want[ok], dt, f"{parts} lens {sl[:6]}")
    if want_lse:  # the suffix pass's OWN log-sum-exp, whatever was merged into `out`
        gl = lse.cpu().numpy()
        fin = np.isfinite(slse)
        assert np.abs(gl[fin] - slse[fin]).max() < 2e-3 and np.all(np.isneginf(gl[~fin]))
