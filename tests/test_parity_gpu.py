"""
-m gpu parity tests proper: the HIP path (through the C ABI) against the committed golden fixtures
(reference Python outputs + float64 oracle) on identical seeded inputs.
"""
import numpy as np
import pytest
import torch

from tests.cases import golden_case_list, make_case
from tests.conftest import load_golden
from tests.gpu_util import assert_close, atol, case_to_device, dev

pytestmark = pytest.mark.gpu

CASES = golden_case_list()


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_hydragen_attention_vs_golden(name, kw):
    from hydragen_amd.attention import hydragen_attention

    case = make_case(**kw)
    g = load_golden(name)
    d = case_to_device(case)
    out = hydragen_attention(**d)
    torch.cuda.synchronize()
    out = out.float().cpu().numpy()
    assert_close(out, g["out_exact"], case["dtype"], f"{name} vs float64 oracle")
    if "out_ref" in g:  # produced by the reference's own python (fp16 cases)
        assert_close(out, g["out_ref"], case["dtype"], f"{name} vs reference python")


@pytest.mark.parametrize("name,kw", [c for c in CASES if make_case(**c[1])["seq_lens"] is not None],
                         ids=[c[0] for c in CASES if make_case(**c[1])["seq_lens"] is not None])
def test_flash_attention_seqlen_vs_golden(name, kw):
    from hydragen_amd.flash import flash_attention_seqlen

    case = make_case(**kw)
    g = load_golden(name)
    d = case_to_device(case)
    for sl in (d["seq_lens"], d["seq_lens"].long()):  # int32 and int64 lengths
        out, lse = flash_attention_seqlen(d["q"], d["k"], d["v"], seq_len=sl)
        torch.cuda.synchronize()
        assert_close(out.float().cpu().numpy(), g["suffix_out_exact"], case["dtype"], f"{name} suffix out")
        lerr = np.abs(lse.cpu().numpy() - g["suffix_lse_exact"]).max()
        assert lerr < 2e-3, f"{name} suffix lse err {lerr}"
        if "suffix_lse_ref" in g:
            assert np.abs(lse.cpu().numpy() - g["suffix_lse_ref"]).max() < 2e-3


def test_combine_lse_vs_reference_grid():
    """tests/test_combine_lse.py grid (bs,seq,heads in 1..3, D in 63/64/128/129, fp32) against the
    reference's combine_lse_torch outputs stored in the fixture."""
    from hydragen_amd.attention import combine_lse

    z = np.load("tests/golden/combine_lse.npz")
    for i in range(int(z["count"])):
        o1, o2, l1, l2 = (dev(z[f"{n}_{i}"]) for n in ("o1", "o2", "l1", "l2"))
        r = combine_lse([o1, o2], [l1, l2]).cpu().numpy()
        assert np.abs(r - z[f"ref_{i}"]).max() < 1e-5, i
    outs = [dev(x) for x in z["n3_outs"]]
    lses = [dev(x) for x in z["n3_lses"]]
    r = combine_lse(outs, lses).cpu().numpy()
    assert np.abs(r - z["n3_ref"]).max() < 1e-5
    # 16-bit dtypes through the vector path
    for dt in ("f16", "bf16"):
        outs = [dev(x, dt) for x in z["n3_outs"]]
        r = combine_lse(outs, lses).float().cpu().numpy()
        from oracle import hydragen_oracle as O
        want = O.combine_lse([o.float().cpu().numpy() for o in outs], list(z["n3_lses"]))
        assert np.abs(r - want).max() <= atol(dt, want)


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_two_stream_form_vs_golden(name, kw):
    """The two-stream form (shared phase on a side stream with persistent prefix workgroups, unique phase beside it,
    log-sum-exp combine after the join; attention.py:250-352 issues the same passes in order) against the same fixtures
    and the same tolerances as the one-call form."""
    from hydragen_amd import attention as A

    case = make_case(**kw)
    g = load_golden(name)
    d = case_to_device(case)
    prev = A.set_two_stream("on")
    cus = A.TWO_STREAM_PREFIX_CUS
    try:
        for A.TWO_STREAM_PREFIX_CUS in (cus, 3):  # 3: far fewer workgroups than units -> every workgroup walks several
            out = A.hydragen_attention(**d)
            torch.cuda.synchronize()
            o = out.float().cpu().numpy()
            assert_close(o, g["out_exact"], case["dtype"], f"{name} two-stream vs float64 oracle")
            if "out_ref" in g:
                assert_close(o, g["out_ref"], case["dtype"], f"{name} two-stream vs reference python")
    finally:
        A.TWO_STREAM_PREFIX_CUS = cus
        A.set_two_stream(prev)


def test_two_stream_form_is_chosen_and_correct_under_graph_capture():
    """Mode 'auto' picks the two-stream form only while a HIP graph is captured and only for shapes with enough prefix work;
    the replayed graph must give the one-call result (up to the unique partial's extra rounding) on every replay, also
    after the inputs changed in place."""
    from hydragen_amd import attention as A

    torch.manual_seed(5)
    B, P, S, H, D = 256, 1024, 32, 32, 128   # 4 * B * H * D * P = 17e9 flops: above the 'auto' threshold
    q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    sk = torch.randn(1, P, H, D, device="cuda", dtype=torch.bfloat16)
    sv = torch.randn_like(sk)
    sl = torch.randint(1, S + 1, (B,), device="cuda", dtype=torch.int32)
    assert not A._want_two_stream(q, k, [sk], [None], [False], capturing=True)  # default mode: off
    prev = A.set_two_stream("auto")
    assert A._want_two_stream(q, k, [sk], [None], [False], capturing=True)
    assert not A._want_two_stream(q, k, [sk], [None], [False], capturing=False)
    assert not A._want_two_stream(q[:2], k[:2], [sk], [None], [False], capturing=True)
    calls = []
    orig = A._launch_decode
    A._launch_decode = lambda lib, p, two, st: (calls.append(two), orig(lib, p, two, st))[1]
    try:
        want = A.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            A.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = A.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
    finally:
        A._launch_decode = orig
        A.set_two_stream(prev)
    assert calls == [False, False, True]
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert (out.float() - want.float()).abs().max().item() <= atol("bf16", want.float()) / 2  # both rounded; the forms differ by one rounding of a 3 % share
    q.copy_(torch.randn_like(q))
    sl.copy_(torch.randint(1, S + 1, (B,), device="cuda", dtype=torch.int32))
    g.replay()
    torch.cuda.synchronize()
    want2 = A.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
    torch.cuda.synchronize()
    assert (out.float() - want2.float()).abs().max().item() <= atol("bf16", want2.float()) / 2


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_f32_partials_vs_golden(name, kw):
    """hyd_decode_params.f32_partials: unsplit levels keep their partial in fp32 (one rounding less than the reference's
    16-bit flash-attn outputs, README.md:488-490); same fixtures, same bounds, one-call and two-stream form."""
    from hydragen_amd import attention as A

    case = make_case(**kw)
    g = load_golden(name)
    d = case_to_device(case)
    prev = A.set_f32_partials(True)
    try:
        for mode in ("off", "on"):
            pm = A.set_two_stream(mode)
            try:
                out = A.hydragen_attention(**d)
                torch.cuda.synchronize()
            finally:
                A.set_two_stream(pm)
            assert_close(out.float().cpu().numpy(), g["out_exact"], case["dtype"], f"{name} f32 partials, two-stream {mode}")
    finally:
        A.set_f32_partials(prev)
