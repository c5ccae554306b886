"""-m gpu: split-KV prefix passes (the C3 / C5-slice shapes: few query rows, long prefix).  The slices of a split pass are
merged at the next launch boundary -- by the suffix kernel's epilogue in the decode operator, by the combine kernel behind
`hyd_prefix_attn_fwd` and in the two-stream form (replaces /root/reference/hydragen/flash.py:76-160 `_splitK_reduce`).  A merge
INSIDE the prefix launch was built and measured in round 5 and is slower on MI355X in every form tried
(profiles/r05_inlaunch_merge_negative.md); these tests were written for it and stay because what they state holds for any
form: the result does not depend on what the caller's workspace held, on the split count, on the form of the operator, or on
how often a captured call is replayed on one workspace."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import _round
from tests.gpu_util import assert_close, dev

pytestmark = pytest.mark.gpu


def _rand(rng, shape, dt):
    return _round(rng.standard_normal(shape, dtype=np.float32), dt)


def _prefix_params(q, k, v, out, lse, num_splits, lse_layout):
    from hydragen_amd._lib import PrefixParams
    from hydragen_amd.flash import _dtype_code

    b, sq, hq, D = q.shape
    p = PrefixParams()
    p.q, p.k, p.v, p.out, p.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr()
    p.k_group_stride, p.k_tok_stride, p.k_head_stride = k.stride(0), k.stride(1), k.stride(2)
    p.v_group_stride, p.v_tok_stride, p.v_head_stride = v.stride(0), v.stride(1), v.stride(2)
    p.dtype = _dtype_code(q)
    p.B, p.nq, p.Hq, p.Hkv, p.D = b, sq, hq, k.shape[2], D
    p.sb, p.kv_len, p.lse_layout, p.num_splits = k.shape[0], k.shape[1], lse_layout, num_splits
    return p


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("fill", ["zeros", "ones", "random"])
def test_workspace_needs_no_initialisation(dt, D, fill):
    """Whatever bytes the caller's workspace holds -- zeros, 0xff (NaNs), noise -- the merged result is the oracle's, call
    after call on the same workspace."""
    from hydragen_amd import _lib
    from hydragen_amd._lib import HYD_LSE_BHQ

    lib = _lib.load()
    rng = np.random.default_rng(D + len(fill))
    b, sq, sk, hq, hkv, splits = 2, 9, 1500, 8, 2, 5
    q, k, v = _rand(rng, (b, sq, hq, D), dt), _rand(rng, (b, sk, hkv, D), dt), _rand(rng, (b, sk, hkv, D), dt)
    tq, tk, tv = dev(q, dt), dev(k, dt), dev(v, dt)
    out = torch.empty_like(tq)
    lse = torch.empty((b, hq, sq), dtype=torch.float32, device=tq.device)
    p = _prefix_params(tq, tk, tv, out, lse, splits, HYD_LSE_BHQ)
    n = lib.hyd_prefix_workspace_bytes(C.byref(p))
    assert n > 0
    if fill == "zeros":
        ws = torch.zeros(n, dtype=torch.uint8, device=tq.device)
    elif fill == "ones":
        ws = torch.full((n,), 0xff, dtype=torch.uint8, device=tq.device)
    elif fill == "random":
        ws = torch.randint(0, 256, (n,), dtype=torch.uint8, device=tq.device)
    else:
        raise AssertionError(fill)
    p.workspace, p.workspace_bytes = ws.data_ptr(), n
    stream = torch.cuda.current_stream().cuda_stream
    want, wlse = O.flash_attention(q, k, v)
    for it in range(3):
        out.zero_()
        lse.zero_()
        _lib.check(lib.hyd_prefix_attn_fwd(C.byref(p), stream))
        torch.cuda.synchronize()
        assert_close(out.float().cpu().numpy(), want, dt, f"split-KV merge, call {it}")
        assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_split_count_does_not_change_the_result(dt):
    """hyd_prefix_attn_fwd on one problem with 8 and with 32 splits (16 units: 128 and 512 workgroups, one and two rounds of the
    chip): both the float64 oracle's answer, and each other's to two units in the last place."""
    from hydragen_amd._lib import HYD_LSE_BQH
    from hydragen_amd.flash import prefix_attention

    rng = np.random.default_rng(3)
    B, P, hq, hkv, D = 64, 4096, 16, 4, 128  # 4 groups x 4 kv heads x 1 row block (16 sequences x 4 heads) = 16 units
    q, k, v = _rand(rng, (B, 1, hq, D), dt), _rand(rng, (4, P, hkv, D), dt), _rand(rng, (4, P, hkv, D), dt)
    tq, tk, tv = dev(q, dt), dev(k, dt), dev(v, dt)
    res = {}
    from hydragen_amd import _lib
    for splits in (8, 32):
        pp = _prefix_params(tq, tk, tv, tq, torch.empty(1, device=tq.device), splits, HYD_LSE_BQH)
        pp.B, pp.nq, pp.sb = B, 1, 4
        ns, grid = C.c_int32(), C.c_int32()
        assert _lib.load().hyd_prefix_plan(C.byref(pp), C.byref(ns), C.byref(grid), None) == 0
        assert (ns.value, grid.value) == (splits, 16 * splits)
        res[splits] = prefix_attention(
            tq, tk, tv, sb=4, kv_len=P, group_stride=(tk.stride(0), tv.stride(0)), tok_stride=(tk.stride(1), tv.stride(1)),
            head_stride=(tk.stride(2), tv.stride(2)), B=B, nq=1, causal=False, lse_layout=HYD_LSE_BQH, lse_shape=(B, 1, hq),
            num_splits=splits)
    torch.cuda.synchronize()
    kk = np.repeat(k, B // 4, axis=0)
    vv = np.repeat(v, B // 4, axis=0)
    want, wlse = O.flash_attention(q, kk, vv)
    for splits, (out, lse) in res.items():
        assert_close(out.float().cpu().numpy(), want, dt, f"{splits} splits")
        assert np.abs(lse.cpu().numpy() - wlse.transpose(0, 2, 1)).max() < 2e-3
    a, b_ = res[8][0].float(), res[32][0].float()
    # different split points round different probabilities to 16 bits: equal to two units in the last place of the largest output
    ulp = 2.0 ** (-7 if dt == "bf16" else -10)
    assert float((a - b_).abs().max()) <= 2 * ulp * float(a.abs().max())


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,P,S,hq,hkv", [(64, 8192, 40, 32, 8), (512, 2048, 17, 8, 1)])
def test_decode_one_call_equals_two_stream_on_split_shapes(dt, B, P, S, hq, hkv):
    """The decode operator on split shapes (C3- and C5-slice-like): the one-call form merges the fp32 slices in the suffix
    kernel's epilogue; the two-stream form runs the prefix pass on 128 persistent workgroups and merges slices + unique
    partial with the combine kernel.  Same answer (16-bit rounding of two fp32 evaluation orders), and the oracle's."""
    from hydragen_amd import attention as A

    rng = np.random.default_rng(B + S)
    D = 128
    q = _rand(rng, (B, 1, hq, D), dt)
    k, v = _rand(rng, (B, S, hkv, D), dt), _rand(rng, (B, S, hkv, D), dt)
    sk, sv = _rand(rng, (1, P, hkv, D), dt), _rand(rng, (1, P, hkv, D), dt)
    sl = rng.integers(1, S + 1, size=B).astype(np.int32)
    args = (dev(q, dt), dev(k, dt), dev(v, dt), [dev(sk, dt)], [dev(sv, dt)], dev(sl))
    one = A.hydragen_attention_nopad(*args)
    prev = A.set_two_stream("on")
    try:
        two = A.hydragen_attention_nopad(*args)
    finally:
        A.set_two_stream(prev)
    torch.cuda.synchronize()
    idx = np.linspace(0, B - 1, 6).astype(int)
    rep = lambda x: np.repeat(x, len(idx), axis=0)  # each picked sequence becomes its own group
    want = O.hydragen_attention_nopad(q[idx], k[idx], v[idx], [rep(sk)], [rep(sv)], sl[idx])
    assert_close(one[idx].float().cpu().numpy(), want, dt, "one-call form")
    assert_close(two[idx].float().cpu().numpy(), want, dt, "two-stream form")
    a, b_ = one.float(), two.float()
    # the two-stream form rounds the unique partial to 16 bits before its merge: allow two units in the last place
    ulp = 2.0 ** (-7 if dt == "bf16" else -10)
    assert float((a - b_).abs().max()) <= 2 * ulp * float(a.abs().max())


def test_graph_replays_on_one_workspace():
    """A captured decode call of a split shape replayed 50 times on one workspace: bit-identical to the eager call."""
    from hydragen_amd import attention as A

    dt = "bf16"
    rng = np.random.default_rng(9)
    B, P, S, hq, hkv, D = 64, 16384, 32, 32, 8, 128
    q = dev(_rand(rng, (B, 1, hq, D), dt), dt)
    k, v = dev(_rand(rng, (B, S, hkv, D), dt), dt), dev(_rand(rng, (B, S, hkv, D), dt), dt)
    sk, sv = dev(_rand(rng, (1, P, hkv, D), dt), dt), dev(_rand(rng, (1, P, hkv, D), dt), dt)
    sl = torch.full((B,), S, dtype=torch.int32, device=q.device)
    eager = A.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = A.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
    for _ in range(50):
        out.zero_()
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
