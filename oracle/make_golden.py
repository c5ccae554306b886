#!/usr/bin/env python3
"""
Generate the committed golden fixtures under tests/golden/ by running the
REFERENCE's own Python (imported from /root/reference, read-only) in this
container, and cross-check the oracle restatement against it.

    TRITON_INTERPRET=1 python oracle/make_golden.py

This script only runs in the authoring container; nothing here travels to the GPU
box except the .npz data it writes.  It is test infrastructure (see
oracle/hydragen_oracle.py header).

What runs for real from the reference:
  * hydragen/attention.py  hydragen_attention / combine_lse_torch / combine_lse_triton
    (+ the Triton kernel combine_lse_kernel under the Triton CPU interpreter)
  * hydragen/flash.py      flash_attention_seqlen, pick_split_k, _splitK_reduce
  * hydragen/xformers_stuff.py  _fwd_kernel_splitK (Triton interpreter, fp16 only:
    numpy has no bf16, so the interpreter cannot run bf16)
What cannot run: the un-vendored third-party CUDA package flash-attn v2.3.6
(requirements.txt:7).  Its two entry points used by the reference
(flash.py:295-304, 336-349) are bound here to exact softmax attention in fp32
(`_flash_attn_forward` / `_flash_attn_varlen_forward` below) -- that is the
published semantics of flash-attn (non-causal / bottom-right causal attention,
natural-log LSE [b,h,sq], output rounded to the input dtype), NOT flash-attn's
code.  Every fixture records which engine produced it.
"""

import json
import os
import sys
import types
from pathlib import Path

os.environ.setdefault("TRITON_INTERPRET", "1")

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REF))

from oracle import hydragen_oracle as O  # noqa: E402
from tests.cases import golden_case_list, make_case  # noqa: E402


# --------------------------------------------------------------------------
# flash-attn boundary: exact attention in fp32 torch (semantics, not code)
# --------------------------------------------------------------------------
def _exact_attn(q, k, v, causal, softmax_scale):
    b, sq, hq, d = q.shape
    sk, hkv = k.shape[1], k.shape[2]
    g = hq // hkv
    qf = q.float().permute(0, 2, 1, 3)  # b h sq d
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(g, 1)
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(g, 1)
    s = torch.matmul(qf, kf.transpose(-1, -2)) * softmax_scale
    if causal:
        i = torch.arange(sq)[:, None]
        j = torch.arange(sk)[None, :]
        s = s.masked_fill(~(j <= i + (sk - sq)), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    o = torch.matmul(p, vf).permute(0, 2, 1, 3).contiguous().to(q.dtype)
    return o, lse.contiguous()


def _flash_attn_forward(q, k, v, dropout_p, causal, softmax_scale, window_size, return_softmax):
    o, lse = _exact_attn(q, k, v, causal, softmax_scale)
    return o, q, k, v, o, lse, None, None


def _flash_attn_varlen_forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                               dropout_p, causal, softmax_scale, window_size, return_softmax):
    n = cu_seqlens_q.shape[0] - 1
    out = torch.zeros_like(q)
    lse = torch.zeros((n, q.shape[1], max_seqlen_q), dtype=torch.float32)
    for i in range(n):
        q0, q1 = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
        k0, k1 = int(cu_seqlens_k[i]), int(cu_seqlens_k[i + 1])
        o, l = _exact_attn(q[None, q0:q1], k[None, k0:k1], v[None, k0:k1], causal, softmax_scale)
        out[q0:q1] = o[0]
        lse[i, :, : q1 - q0] = l[0]
    return out, q, k, v, out, lse, None, None


def install_shims():
    fa = types.ModuleType("flash_attn")
    fai = types.ModuleType("flash_attn.flash_attn_interface")
    fai._flash_attn_forward = _flash_attn_forward
    fai._flash_attn_varlen_forward = _flash_attn_varlen_forward
    fa.flash_attn_interface = fai
    sys.modules["flash_attn"] = fa
    sys.modules["flash_attn.flash_attn_interface"] = fai

    # flash.py:193 reads the SM count of a CUDA device
    class _Props:
        multi_processor_count = 108

    torch.cuda.get_device_properties = lambda *_a, **_k: _Props()

    import hydragen.xformers_stuff as xs

    # Triton 3.6's inspect path needs a list of lines, the original returns a str
    def _getlines(filename, module_globals=None):
        if filename in xs._FILENAME_TO_SRC:
            return xs._FILENAME_TO_SRC[filename].splitlines(keepends=True)
        return xs._getlines_orig(filename, module_globals)

    xs._monkey_patched_getlines = _getlines


def t16(x, dtype):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.float16 if dtype == "f16" else torch.bfloat16)


def run_reference_case(case, ref_attn, ref_flash):
    dt = case["dtype"]
    q, k, v = (t16(case[n], dt) for n in "qkv")
    sks = [t16(x, dt) for x in case["shared_ks"]]
    svs = [t16(x, dt) for x in case["shared_vs"]]
    cus = [None if c is None else torch.from_numpy(c) for c in case["shared_cu_seq_lens"]]
    sl = None if case["seq_lens"] is None else torch.from_numpy(case["seq_lens"])
    out = ref_attn.hydragen_attention(
        q=q, k=k, v=v, shared_ks=sks, shared_vs=svs, shared_cu_seq_lens=cus,
        shared_max_seq_lens=case["shared_max_seq_lens"], use_varlens=case["use_varlens"], seq_lens=sl,
    )
    extra = {}
    if sl is not None:
        so, slse = ref_flash.flash_attention_seqlen(q, k, v, seq_len=sl)
        extra["suffix_out_ref"] = so.float().numpy()
        extra["suffix_lse_ref"] = slse.float().numpy()
    return out.float().numpy(), extra


def main():
    install_shims()
    import hydragen.attention as ref_attn
    import hydragen.flash as ref_flash

    outdir = REPO / "tests" / "golden"
    outdir.mkdir(parents=True, exist_ok=True)
    report = []

    for name, kw in golden_case_list():
        case = make_case(**kw)
        dt = case["dtype"]
        exact = O.hydragen_attention(
            case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"],
            case["shared_cu_seq_lens"], case["shared_max_seq_lens"], case["use_varlens"], case["seq_lens"],
        )
        nosh = O.nosharing_attention(
            case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"],
            case["shared_cu_seq_lens"], case["use_varlens"], case["seq_lens"],
        )
        # decomposition identity of the oracle itself
        assert np.abs(exact - nosh).max() < 1e-9, name
        save = dict(out_exact=exact.astype(np.float32))
        if case["seq_lens"] is not None:
            so, sl = O.flash_attention_seqlen(case["q"], case["k"], case["v"], case["seq_lens"])
            save["suffix_out_exact"] = so.astype(np.float32)
            save["suffix_lse_exact"] = sl.astype(np.float32)
        if dt == "f16":
            ref_out, extra = run_reference_case(case, ref_attn, ref_flash)
            engine = ("reference python (attention.py, flash.py, xformers_stuff.py; Triton kernels under "
                      "TRITON_INTERPRET=1) + exact fp32 softmax attention at the flash-attn 2.3.6 boundary")
            save["out_ref"] = ref_out.astype(np.float32)
            save.update(extra)
            err = np.abs(ref_out - exact).max()
            mrd = O.rdiff(ref_out, exact).mean()
            # the reference's own acceptance bar (tests/test_attention.py:36-38,185)
            assert err <= 2e-3 and mrd <= 5e-3, (name, err, mrd)
            if "suffix_lse_ref" in extra:
                lerr = np.abs(extra["suffix_lse_ref"] - save["suffix_lse_exact"]).max()
                assert lerr < 1e-4, (name, lerr)
            report.append((name, engine.split(";")[0], float(err), float(mrd)))
        else:
            engine = "oracle restatement (float64 maths on bf16-rounded inputs); bf16 cannot run under the Triton interpreter"
            report.append((name, "oracle", 0.0, 0.0))
        meta = dict(kw)
        meta["engine"] = engine
        save["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(outdir / f"{name}.npz", **save)

    # ---- combine_lse grid (tests/test_combine_lse.py:11-14) against combine_lse_torch ----
    rng = np.random.default_rng(2024)
    comb = {}
    idx = 0
    for bs in (1, 2, 3):
        for sl in (1, 2, 3):
            for h in (1, 2, 3):
                for d in (63, 64, 128, 129):
                    o1, o2 = rng.random((2, bs, sl, h, d), dtype=np.float32)
                    l1, l2 = rng.random((2, bs, sl, h), dtype=np.float32)
                    r = ref_attn.combine_lse_torch([torch.from_numpy(o1), torch.from_numpy(o2)],
                                                   [torch.from_numpy(l1), torch.from_numpy(l2)]).numpy()
                    mine = O.combine_lse([o1, o2], [l1, l2])
                    assert np.abs(r - mine).max() < 1e-6
                    if d in (64, 128):  # the Triton kernel is only correct for D % 64 == 0
                        rt = ref_attn.combine_lse_triton(torch.from_numpy(o1), torch.from_numpy(l1),
                                                         torch.from_numpy(o2), torch.from_numpy(l2)).numpy()
                        assert np.abs(rt - mine).max() < 1e-5
                    comb[f"o1_{idx}"], comb[f"o2_{idx}"] = o1, o2
                    comb[f"l1_{idx}"], comb[f"l2_{idx}"] = l1, l2
                    comb[f"ref_{idx}"] = r
                    idx += 1
    # 3-partial combine (C4-shaped, tiny) -> combine_lse_torch (attention.py:169-174 picks it for N != 2)
    o = rng.standard_normal((3, 4, 1, 8, 128)).astype(np.float32)
    l = (rng.standard_normal((3, 4, 1, 8)) * 3).astype(np.float32)
    r3 = ref_attn.combine_lse_torch([torch.from_numpy(x) for x in o], [torch.from_numpy(x) for x in l]).numpy()
    assert np.abs(r3 - O.combine_lse(list(o), list(l))).max() < 1e-5
    comb["n3_outs"], comb["n3_lses"], comb["n3_ref"] = o, l, r3
    comb["count"] = np.asarray(idx)
    np.savez_compressed(outdir / "combine_lse.npz", **comb)

    for r in report:
        print("%-28s %-40s max|ref-oracle|=%.2e mean rdiff=%.2e" % r)
    print("wrote", len(report) + 1, "fixtures to", outdir)


if __name__ == "__main__":
    main()
