"""-m "not gpu": pin the oracles (numpy float64, plain C, torch-CPU port) against the committed
golden fixtures, whose fp16 entries were produced by the reference's own Python
(oracle/make_golden.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import golden_case_list, make_case
from tests.conftest import load_golden

CASES = golden_case_list()
IDS = [c[0] for c in CASES]


def _call_oracle(case, **kw):
    return O.hydragen_attention(case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"],
                                case["shared_cu_seq_lens"], case["shared_max_seq_lens"], case["use_varlens"],
                                case["seq_lens"], **kw)


@pytest.mark.parametrize("name,kw", CASES, ids=IDS)
def test_numpy_oracle_vs_golden(name, kw):
    case = make_case(**kw)
    g = load_golden(name)
    out = _call_oracle(case)
    assert np.abs(out - g["out_exact"]).max() < 1e-6
    if "out_ref" in g:
        # the reference's own acceptance bar (tests/test_attention.py:36-38,182-187)
        assert np.abs(out - g["out_ref"]).max() <= 2e-3
        assert O.rdiff(out, g["out_ref"]).mean() <= 5e-3
        assert "reference python" in g["meta"]["engine"]
    # decomposed == undecomposed (what the reference test itself asserts)
    nosh = O.nosharing_attention(case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"],
                                 case["shared_cu_seq_lens"], case["use_varlens"], case["seq_lens"])
    assert np.abs(out - nosh).max() < 1e-9


@pytest.mark.parametrize("name,kw", [c for c in CASES if "ref_spec4" not in c[0]], ids=[i for i in IDS if "ref_spec4" not in i])
def test_c_oracle_suffix_vs_reference_kernels(name, kw, oracle_so):
    """The kernel-level C restatement (base-2 online softmax, p rounded to the q dtype, split-K reduce)
    against what the reference's Triton kernels produced under the interpreter."""
    case = make_case(**kw)
    g = load_golden(name)
    if case["seq_lens"] is None:
        pytest.skip("no ragged suffix in this case")
    dt = case["dtype"]
    code = 0 if dt == "f16" else 1
    q, k, v = (np.ascontiguousarray(O.to_bits(case[n], dt)) for n in "qkv")
    B, nq, Hq, D = case["q"].shape
    Mk, Hkv = case["k"].shape[1], case["k"].shape[2]
    out = np.zeros((B, nq, Hq, D), dtype=np.float32)
    lse = np.zeros((B, nq, Hq), dtype=np.float32)
    sl = np.ascontiguousarray(case["seq_lens"], dtype=np.int32)
    oracle_so.orc_pick_split_k.restype = C.c_int
    g_ = Hq // Hkv
    M = nq * g_
    block_m = max(16, min(1 << (M - 1).bit_length(), 128))
    split_k = oracle_so.orc_pick_split_k(B, 1, M, block_m, Mk, 64, 108)  # flash.py:188-196 (H=1, stubbed 108 SMs)
    oracle_so.orc_suffix_splitk(
        q.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), code,
        B, nq, Hq, Hkv, D, Mk, sl.ctypes.data_as(C.c_void_p), split_k,
        out.ctypes.data_as(C.c_void_p), lse.ctypes.data_as(C.c_void_p))
    assert np.abs(lse - g["suffix_lse_exact"]).max() < 1e-4
    tol = 2e-3 if dt == "f16" else 1.6e-2
    assert np.abs(out - g["suffix_out_exact"]).max() <= tol
    if "suffix_out_ref" in g:
        # same arithmetic order as the Triton kernels -> agreement to ~1 fp16 ulp
        assert np.abs(out - g["suffix_out_ref"]).max() <= 1e-3
        assert np.abs(lse - g["suffix_lse_ref"]).max() < 1e-5
        assert (out == g["suffix_out_ref"]).mean() > 0.97


@pytest.mark.parametrize("name,kw", [c for c in CASES if c[0] in ("c1_literal_f16", "c1_literal_bf16", "two_level_bf16", "ragged_g4_f16")],
                         ids=["c1_literal_f16", "c1_literal_bf16", "ragged_g4_f16", "two_level_bf16"])
def test_c_oracle_decode_and_torch_port(name, kw, oracle_so):
    case = make_case(**kw)
    g = load_golden(name)
    dt = case["dtype"]
    # torch CPU port (bench.py's cpu_baseline)
    from oracle import cpu_port_torch as port
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    out = port.hydragen_attention_nopad(t(case["q"]), t(case["k"]), t(case["v"]), [t(x) for x in case["shared_ks"]],
                                        [t(x) for x in case["shared_vs"]], t(case["seq_lens"]).long()).numpy()
    assert np.abs(out - g["out_exact"]).max() < 2e-5
    if len(case["shared_ks"]) == 1:
        ns = port.nosharing_attention(t(case["q"]), t(case["k"]), t(case["v"]), t(case["shared_ks"][0]),
                                      t(case["shared_vs"][0]), t(case["seq_lens"]).long()).numpy()
        assert np.abs(ns - g["out_exact"]).max() < 2e-5
        # plain-C decode restatement (partials rounded to the q dtype like the reference)
        code = 0 if dt == "f16" else 1
        q, k, v = (np.ascontiguousarray(O.to_bits(case[n], dt)) for n in "qkv")
        sk = np.ascontiguousarray(O.to_bits(case["shared_ks"][0], dt))
        sv = np.ascontiguousarray(O.to_bits(case["shared_vs"][0], dt))
        B, nq, Hq, D = case["q"].shape
        Mk, Hkv = case["k"].shape[1], case["k"].shape[2]
        sb, P = case["shared_ks"][0].shape[:2]
        n = B * nq * Hq
        scratch = np.zeros(2 * n * D + 2 * n, dtype=np.float32)
        res = np.zeros((B, nq, Hq, D), dtype=np.float32)
        sl = np.ascontiguousarray(case["seq_lens"], dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        oracle_so.orc_hydragen_decode(p(q), p(k), p(v), p(sk), p(sv), code, B, nq, Hq, Hkv, D, sb, P, Mk, p(sl),
                                      p(scratch), p(res))
        tol = 2e-3 if dt == "f16" else 1.6e-2
        assert np.abs(res - g["out_exact"]).max() <= tol
        if "out_ref" in g:
            assert np.abs(res - g["out_ref"]).max() <= 1e-3


def test_combine_fixture_vs_oracle():
    z = np.load("tests/golden/combine_lse.npz")
    for i in range(int(z["count"])):
        r = O.combine_lse([z[f"o1_{i}"], z[f"o2_{i}"]], [z[f"l1_{i}"], z[f"l2_{i}"]])
        assert np.abs(r - z[f"ref_{i}"]).max() < 1e-6
    r = O.combine_lse(list(z["n3_outs"]), list(z["n3_lses"]))
    assert np.abs(r - z["n3_ref"]).max() < 1e-5


def test_rounding_helpers(oracle_so):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4096) * np.exp(rng.uniform(-12, 12, 4096))).astype(np.float32)
    x[:4] = [0.0, -0.0, 65504.0, 1e-8]
    for dt, code in (("f16", 0), ("bf16", 1)):
        bits = np.zeros(x.size, dtype=np.uint16)
        oracle_so.orc_round_array(x.ctypes.data_as(C.c_void_p), bits.ctypes.data_as(C.c_void_p), C.c_long(x.size), code)
        assert np.array_equal(bits, O.to_bits(x, dt))
        if dt == "bf16":
            t = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
            assert np.array_equal(bits, t)
