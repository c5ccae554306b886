#!/usr/bin/env python3
"""Development probe (ablation library): the matrix-core suffix kernel at reduced occupancy (HYD_GQA_LDS_PAD) on MHA
and grouped-query shapes -- would it stream at 4 waves per CU, i.e. as a role of the co-run kernel?"""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hydragen_amd import _lib
from hydragen_amd._lib import SuffixParams
from hydragen_amd.flash import fill_suffix_params
lib = _lib.load(); dev = "cuda:0"; dt = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (B, Hq, Hkv, S) in ((1024, 32, 32, 64), (1024, 32, 32, 128), (2048, 8, 1, 256), (1024, 32, 8, 128)):
    q = torch.randn(B, 1, Hq, 128, device=dev, dtype=dt)
    k = torch.randn(B, S, Hkv, 128, device=dev, dtype=dt); v = torch.randn_like(k)
    out = torch.empty_like(q); lse = torch.empty(B, 1, Hq, device=dev, dtype=torch.float32)
    sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    sp = SuffixParams(); fill_suffix_params(sp, q, k, v, sl, out); sp.lse = lse.data_ptr()
    call = lambda: _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), stream))
    by = 2 * 2 * Hkv * 128 * B * S + 2 * B * Hq * 128 * 2
    os.environ.pop("HYD_GQA_LDS_PAD", None)
    line = f"B={B} {Hq}/{Hkv} S={S}:"
    for pad, label in ((None, "10/CU"), (24 * 1024, "4/CU"), (37 * 1024, "3/CU"), (60 * 1024, "2/CU")):
        if pad is None: os.environ.pop("HYD_GQA_LDS_PAD", None)
        else: os.environ["HYD_GQA_LDS_PAD"] = str(pad)
        t = timeit(call)
        line += f"  {label} {t:7.1f} us ({by / t / 1e6:4.2f} TB/s)"
    print(line, flush=True)
