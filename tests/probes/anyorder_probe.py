#!/usr/bin/env python3
"""Probe (ablation library): does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let the suffix pass start
beside the prefix pass in ONE queue?  Timing only -- the suffix epilogue then races with the prefix partial.

    HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so python tests/probes/anyorder_probe.py
"""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

from hydragen_amd import _lib
from hydragen_amd._lib import DecodeParams
from hydragen_amd.attention import _fill_level
from hydragen_amd.flash import fill_suffix_params

lib = _lib.load()
dev = "cuda:0"
B, P, Smax, H, D = 1024, 2048, 128, 32, 128
dt = torch.bfloat16
q = torch.randn(B, 1, H, D, device=dev, dtype=dt)
k = torch.randn(B, Smax, H, D, device=dev, dtype=dt)
v = torch.randn_like(k)
sk = torch.randn(1, P, H, D, device=dev, dtype=dt)
sv = torch.randn_like(sk)
out = torch.empty_like(q)
stream = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for s in (16, 64, 128):
    sl = torch.full((B,), s, dtype=torch.int32, device=dev)
    p = DecodeParams()
    fill_suffix_params(p.suffix, q, k, v, sl, out)
    p.n_levels = 1
    _fill_level(p.levels[0], sk, sv, None, None, False, B)
    n = lib.hyd_decode_workspace_bytes(C.byref(p))
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), n
    call = lambda: _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))
    line = f"S={s:4d}:"
    for np_ in (0, 64, 96, 128, 192):
        for ao in (0, 1):
            os.environ["HYD_ANYORDER"] = str(ao)
            if np_:
                os.environ["HYD_PREFIX_PERSIST"] = str(np_)
            else:
                os.environ.pop("HYD_PREFIX_PERSIST", None)
            line += f"  Np={np_}/{'any' if ao else 'ord'} {timeit(call):6.1f}"
    print(line, flush=True)
