#!/usr/bin/env python3
"""Randomized soak of the whole operator against the float64 oracle (development tool; the generator is the one of
tests/test_fuzz_gpu.py with other seeds):   python tools/soak.py --first 1000 --count 1500"""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oracle import hydragen_oracle as O
from tests.cases import make_case
from tests.gpu_util import case_to_device, rdiff
from tests.test_fuzz_gpu import _draw, check_fuzz
from hydragen_amd.attention import hydragen_attention

ap = argparse.ArgumentParser(); ap.add_argument("--first", type=int, default=1000); ap.add_argument("--count", type=int, default=500); ap.add_argument("--dim", type=int, default=0, help="force this head dim (e.g. 256)"); ap.add_argument("--two-stream", action="store_true"); ap.add_argument("--f32-partials", action="store_true")
a = ap.parse_args()
import numpy as np
from hydragen_amd import attention as A
if a.two_stream: A.set_two_stream('on')
if a.f32_partials: A.set_f32_partials(True)
bad = 0; worst = {"f16": 0.0, "bf16": 0.0}; worst_l2 = {"f16": 0.0, "bf16": 0.0}; worst_abs = {"f16": 0.0, "bf16": 0.0}
for seed in range(a.first, a.first + a.count):
    kw, prefill = _draw(seed)
    if a.dim: kw['dim'] = a.dim
    case = make_case(**kw)
    if prefill: case["seq_lens"] = None
    out = hydragen_attention(**case_to_device(case)); torch.cuda.synchronize()
    want = O.hydragen_attention(case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"], case["shared_cu_seq_lens"],
                                case["shared_max_seq_lens"], case["use_varlens"], case["seq_lens"])
    g_ = out.float().cpu().numpy().astype(np.float64)
    l2 = float(np.linalg.norm(g_ - want) / max(np.linalg.norm(want), 1e-30))
    worst_l2[case["dtype"]] = max(worst_l2[case["dtype"]], l2)
    worst[case["dtype"]] = max(worst[case["dtype"]], float(rdiff(g_, want).mean()))
    worst_abs[case["dtype"]] = max(worst_abs[case["dtype"]], float(np.abs(g_ - want).max()))
    try:
        check_fuzz(out.float().cpu().numpy(), want, case["dtype"], f"seed {seed}")
    except AssertionError as e:
        bad += 1; print("FAIL", seed, kw["sizes"], kw["qheads"], kw["kvheads"], kw["dim"], kw["dtype"], str(e)[:120], flush=True)
print(f"{a.count} cases, {bad} failures, worst mean rdiff {worst}, worst relative L2 error {worst_l2}, worst abs {worst_abs}")
