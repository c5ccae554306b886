"""-m gpu: the attention primitives (reference flash.py signatures) against the float64 oracle."""
import zlib

import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import _round
from tests.gpu_util import assert_close, dev

pytestmark = pytest.mark.gpu


def _rand(rng, shape, dt):
    return _round(rng.standard_normal(shape, dtype=np.float32), dt)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("b,sq,sk,hq,hkv,causal", [
    (1, 1, 1, 4, 4, False),          # single key
    (2, 5, 37, 8, 2, False),         # GQA, ragged tile tail
    (1, 300, 129, 4, 1, False),      # > 128 folded rows per kv head (3 row blocks), 2 key tiles
    (3, 16, 16, 4, 4, True),         # causal square (unique-suffix prefill shape)
    (2, 7, 200, 8, 4, True),         # causal, bottom-right aligned (sq < sk)
    (1, 130, 130, 2, 1, True),       # causal across row blocks with GQA fold
])
def test_flash_attention(dt, D, b, sq, sk, hq, hkv, causal):
    from hydragen_amd.flash import flash_attention

    rng = np.random.default_rng(zlib.crc32(repr((dt, D, b, sq, sk, hq, hkv, causal)).encode()))
    q, k, v = _rand(rng, (b, sq, hq, D), dt), _rand(rng, (b, sk, hkv, D), dt), _rand(rng, (b, sk, hkv, D), dt)
    out, lse = flash_attention(dev(q, dt), dev(k, dt), dev(v, dt), causal=causal)
    torch.cuda.synchronize()
    want, wlse = O.flash_attention(q, k, v, causal=causal)
    assert lse.shape == (b, hq, sq) and lse.dtype == torch.float32
    assert_close(out.float().cpu().numpy(), want, dt, "flash_attention out")
    assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("splits", [2, 3, 7])
def test_flash_attention_split_kv(dt, splits):
    """Forced split-KV (the C3-shape path): slices + in-library merge must equal the unsplit answer."""
    from hydragen_amd.flash import prefix_attention
    from hydragen_amd._lib import HYD_LSE_BHQ

    rng = np.random.default_rng(splits)
    b, sq, sk, hq, hkv, D = 2, 9, 1000, 8, 2, 128
    q, k, v = _rand(rng, (b, sq, hq, D), dt), _rand(rng, (b, sk, hkv, D), dt), _rand(rng, (b, sk, hkv, D), dt)
    tq, tk, tv = dev(q, dt), dev(k, dt), dev(v, dt)
    out, lse = prefix_attention(
        tq, tk, tv, sb=b, kv_len=sk, group_stride=(tk.stride(0), tv.stride(0)),
        tok_stride=(tk.stride(1), tv.stride(1)), head_stride=(tk.stride(2), tv.stride(2)),
        B=b, nq=sq, causal=False, lse_layout=HYD_LSE_BHQ, lse_shape=(b, hq, sq), num_splits=splits)
    torch.cuda.synchronize()
    want, wlse = O.flash_attention(q, k, v)
    assert_close(out.float().cpu().numpy(), want, dt, "split-kv out")
    assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_flash_attention_varlen(dt):
    from hydragen_amd.flash import flash_attention_varlen

    rng = np.random.default_rng(5)
    hq, hkv, D = 8, 2, 128
    for causal in (False, True):
        # causal = bottom-right aligned (flash.py:336-349 passes it through): every query must keep at least one key
        qlens, klens = ([3, 1, 7, 140], [9, 130, 1, 64]) if not causal else ([3, 1, 7, 140], [9, 130, 7, 200])
        cu_q = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
        cu_k = np.concatenate([[0], np.cumsum(klens)]).astype(np.int32)
        q, k, v = _rand(rng, (cu_q[-1], hq, D), dt), _rand(rng, (cu_k[-1], hkv, D), dt), _rand(rng, (cu_k[-1], hkv, D), dt)
        out, lse = flash_attention_varlen(dev(q, dt), dev(k, dt), dev(v, dt), dev(cu_q), dev(cu_k),
                                          max(qlens), max(klens), causal=causal)
        torch.cuda.synchronize()
        want, wlse = O.flash_attention_varlen(q, k, v, cu_q, cu_k, max(qlens), max(klens), causal=causal)
        assert_close(out.float().cpu().numpy(), want, dt, "varlen out")
        assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("b,nq,mk,hq,hkv,lens", [
    (4, 1, 16, 8, 8, [5, 16, 3, 1]),
    (3, 1, 300, 8, 1, [300, 1, 77]),     # g = 8 rows per kv head; few units -> in-workgroup key split
    (2, 3, 40, 4, 2, [40, 17]),          # several queries per sequence (nq*g = 6 rows)
    (2, 5, 33, 8, 2, [33, 2]),           # nq*g = 20 rows -> row chunks
    (5, 1, 8, 4, 4, [0, 8, 1, 0, 3]),    # empty sequences
    (3, 1, 70, 12, 4, [70, 1, 33]),      # g = 3: the smallest row count on the matrix-core kernel
    (300, 1, 40, 6, 2, None),            # g = 3, many units: one wave per unit, 2 kv heads per workgroup
    (300, 1, 40, 40, 8, None),           # g = 5, 8 kv heads: 4 of them per workgroup
    (260, 1, 40, 64, 16, None),          # 16 kv heads: one per workgroup
])
def test_flash_attention_seqlen(dt, D, b, nq, mk, hq, hkv, lens):
    from hydragen_amd.flash import flash_attention_seqlen

    rng = np.random.default_rng(zlib.crc32(repr((dt, D, b, nq, mk)).encode()))
    q, k, v = _rand(rng, (b, nq, hq, D), dt), _rand(rng, (b, mk, hkv, D), dt), _rand(rng, (b, mk, hkv, D), dt)
    sl = np.asarray(lens, dtype=np.int32) if lens is not None else rng.integers(0, mk + 1, b).astype(np.int32)
    out, lse = flash_attention_seqlen(dev(q, dt), dev(k, dt), dev(v, dt), seq_len=dev(sl))
    torch.cuda.synchronize()
    want, wlse = O.flash_attention_seqlen(q, k, v, sl)
    assert_close(out.float().cpu().numpy(), want, dt, "seqlen out")
    got_lse = lse.cpu().numpy()
    fin = np.isfinite(wlse)
    assert np.abs(got_lse[fin] - wlse[fin]).max() < 2e-3
    assert np.all(np.isneginf(got_lse[~fin]))


def test_unique_prefill_causal_suffix():
    """seq_lens=None with nq == kvlen: the reference attends the unique part causally
    (attention.py:343-345) -- UNIQUE_PREFILL shape, two shared levels, N-way merge."""
    from hydragen_amd.attention import hydragen_attention_nopad

    dt = "f16"
    rng = np.random.default_rng(11)
    B, nq, hq, hkv, D = 4, 6, 8, 2, 128
    q = _rand(rng, (B, nq, hq, D), dt)
    k, v = _rand(rng, (B, nq, hkv, D), dt), _rand(rng, (B, nq, hkv, D), dt)
    sks = [_rand(rng, (1, 50, hkv, D), dt), _rand(rng, (2, 20, hkv, D), dt)]
    svs = [_rand(rng, (1, 50, hkv, D), dt), _rand(rng, (2, 20, hkv, D), dt)]
    out = hydragen_attention_nopad(dev(q, dt), dev(k, dt), dev(v, dt), [dev(x, dt) for x in sks],
                                   [dev(x, dt) for x in svs], seq_len=None)
    torch.cuda.synchronize()
    want = O.hydragen_attention_nopad(q, k, v, sks, svs, None)
    assert_close(out.float().cpu().numpy(), want, dt, "unique prefill")


def test_prefix_only_early_exit():
    """attention.py:273-274: empty unique KV and one level -> the prefix result."""
    from hydragen_amd.attention import hydragen_attention_nopad

    dt = "bf16"
    rng = np.random.default_rng(12)
    B, hq, hkv, D = 8, 8, 8, 128
    q = _rand(rng, (B, 1, hq, D), dt)
    sk, sv = _rand(rng, (2, 70, hkv, D), dt), _rand(rng, (2, 70, hkv, D), dt)
    k = torch.empty(B, 0, hkv, D, dtype=torch.bfloat16, device="cuda:0")
    out = hydragen_attention_nopad(dev(q, dt), k, k.clone(), [dev(sk, dt)], [dev(sv, dt)])
    torch.cuda.synchronize()
    want = O.hydragen_attention_nopad(q, np.zeros((B, 0, hkv, D)), np.zeros((B, 0, hkv, D)), [sk], [sv])
    assert_close(out.float().cpu().numpy(), want, dt, "prefix only")


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("B,P", [(4, 1024), (2, 700), (16, 513)])
def test_prefix_only_early_exit_with_split_kv(dt, B, P):
    """the early exit of attention.py:273-274 on shapes whose prefix pass splits the keys (few units, long
    prefix) -- the workspace query and the launch must agree."""
    from hydragen_amd.attention import hydragen_attention_nopad

    rng = np.random.default_rng(B * 1000 + P)
    hq, hkv, D = 8, 8, 128
    q = _rand(rng, (B, 1, hq, D), dt)
    sk, sv = _rand(rng, (1, P, hkv, D), dt), _rand(rng, (1, P, hkv, D), dt)
    k = torch.empty(B, 0, hkv, D, dtype=dev(q, dt).dtype, device="cuda:0")
    out = hydragen_attention_nopad(dev(q, dt), k, k.clone(), [dev(sk, dt)], [dev(sv, dt)])
    torch.cuda.synchronize()
    want = O.hydragen_attention_nopad(q, np.zeros((B, 0, hkv, D)), np.zeros((B, 0, hkv, D)), [sk], [sv])
    assert_close(out.float().cpu().numpy(), want, dt, "prefix only, split-KV")


def test_deep_hierarchy_of_long_levels_single_kv_head():
    """three long shared levels with one kv head (the TP=8 70B shard's head layout): every level wants the
    maximum split; together they must still fit the epilogue's merge budget and give the oracle's answer."""
    from hydragen_amd.attention import hydragen_attention_nopad

    dt, rng = "bf16", np.random.default_rng(77)
    B, hq, hkv, D, n = 16, 8, 1, 128, 8192
    q = _rand(rng, (B, 1, hq, D), dt)
    k, v = _rand(rng, (B, 5, hkv, D), dt), _rand(rng, (B, 5, hkv, D), dt)
    sks = [_rand(rng, (sb, n, hkv, D), dt) for sb in (1, 2, 4)]
    svs = [_rand(rng, (sb, n, hkv, D), dt) for sb in (1, 2, 4)]
    lens = np.asarray([5, 1, 3, 2] * 4, dtype=np.int32)
    out = hydragen_attention_nopad(dev(q, dt), dev(k, dt), dev(v, dt), [dev(x, dt) for x in sks],
                                   [dev(x, dt) for x in svs], dev(lens))
    torch.cuda.synchronize()
    want = O.hydragen_attention_nopad(q, k, v, sks, svs, lens)
    assert_close(out.float().cpu().numpy(), want, dt, "3 long levels, 1 kv head")


@pytest.mark.parametrize("levels", [[(1, 300)], [(1, 200), (4, 40)], [(1, 4096)]])
def test_decode_phases_equal_the_one_call_form(levels):
    """HYD_PHASE_SHARED followed by HYD_PHASE_UNIQUE (what bench.py issues to put an event between the two kernels)
    is bit-identical to HYD_PHASE_ALL."""
    import ctypes as C
    from hydragen_amd import _lib
    from hydragen_amd._lib import DecodeParams
    from hydragen_amd.attention import _fill_level
    from hydragen_amd.flash import fill_suffix_params

    dt, rng = "bf16", np.random.default_rng(len(levels) * 31 + levels[0][1])
    B, hq, hkv, D, S = 8, 8, 2, 128, 12
    q = dev(_rand(rng, (B, 1, hq, D), dt), dt)
    k, v = dev(_rand(rng, (B, S, hkv, D), dt), dt), dev(_rand(rng, (B, S, hkv, D), dt), dt)
    sks = [dev(_rand(rng, (sb, n, hkv, D), dt), dt) for sb, n in levels]
    svs = [dev(_rand(rng, (sb, n, hkv, D), dt), dt) for sb, n in levels]
    lens = dev(np.asarray([12, 1, 5, 7, 12, 3, 9, 2], dtype=np.int32))
    lib = _lib.load()
    outs = []
    for phases in ((0,), (1, 2)):
        out = torch.zeros_like(q)
        ps = []
        for ph in phases:
            p = DecodeParams()
            fill_suffix_params(p.suffix, q, k, v, lens, out)
            p.n_levels, p.phase = len(levels), ph
            for i, (sk, sv) in enumerate(zip(sks, svs)):
                _fill_level(p.levels[i], sk, sv, None, None, False, B)
            ps.append(p)
        nbytes = lib.hyd_decode_workspace_bytes(C.byref(ps[0]))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda:0")
        for p in ps:
            p.workspace, p.workspace_bytes = ws.data_ptr(), nbytes
            _lib.check(lib.hyd_decode_attn_fused(C.byref(p), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert outs[0].abs().sum().item() > 0


def test_misaligned_views_are_accepted():
    """a contiguous tensor at a storage offset that is not a multiple of 16 bytes (the reference takes any
    view) is copied to aligned memory instead of being refused."""
    from hydragen_amd.flash import flash_attention_seqlen

    dt, rng = "f16", np.random.default_rng(3)
    B, S, H, D = 3, 9, 4, 64
    q, k, v = _rand(rng, (B, 1, H, D), dt), _rand(rng, (B, S, H, D), dt), _rand(rng, (B, S, H, D), dt)

    def off(x):
        flat = torch.empty(x.size + 4, dtype=torch.float16, device="cuda:0")
        flat[4:] = dev(x, dt).reshape(-1)
        t = flat[4:].view(x.shape)
        assert t.data_ptr() % 16 != 0 and t.is_contiguous()
        return t

    out, _ = flash_attention_seqlen(off(q), off(k), off(v), seq_len=None)
    torch.cuda.synchronize()
    want, _ = O.flash_attention_seqlen(q, k, v, None)
    assert_close(out.float().cpu().numpy(), want, dt, "misaligned views")
