"""
-m gpu parity tests proper: the HIP path (through the C ABI) against the committed golden fixtures
(reference Python outputs + float64 oracle) on identical seeded inputs.
"""
import numpy as np
import pytest
import torch

from tests.cases import golden_case_list, make_case
from tests.conftest import load_golden
from tests.gpu_util import ATOL, assert_close, case_to_device, dev

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
        assert np.abs(r - want).max() <= ATOL[dt]
