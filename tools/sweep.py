#!/usr/bin/env python3
"""Shape sweep of the two passes (development tool): the (B, P) grid of the paper's microbenchmark
(docs/sweeps_from_paper.md:159-161) for the reference's default heads (8/1) and for MHA 32/32; prints the prefix
pass in TFLOP/s and the fused suffix pass in GB/s so that shape-dependent cliffs show up."""
import argparse, sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hydragen_amd import _lib
from hydragen_amd._lib import PrefixParams, SuffixParams, HYD_LSE_BQH
from hydragen_amd.flash import fill_suffix_params, _dtype_code

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=128)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
lib = _lib.load(); dev = "cuda:0"; dt = torch.bfloat16; D = 128
stream = torch.cuda.current_stream().cuda_stream

def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

print(f"{'heads':>6s} {'B':>5s} {'P':>6s} | {'prefix us':>9s} {'TF/s':>7s} {'splits':>6s} | {'suffix us (S=%d)' % a.S:>17s} {'GB/s':>7s}")
for Hq, Hkv in ((8, 1), (32, 32), (32, 8)):
    for B in (16, 64, 256, 1024, 2048):
        q = torch.randn(B, 1, Hq, D, device=dev, dtype=dt)
        k = torch.randn(B, a.S, Hkv, D, device=dev, dtype=dt); v = torch.randn_like(k)
        out = torch.empty_like(q); pout = torch.randn_like(q); plse = torch.randn(B, 1, Hq, device=dev, dtype=torch.float32)
        sl = torch.full((B,), a.S, dtype=torch.int32, device=dev)
        sp = SuffixParams(); fill_suffix_params(sp, q, k, v, sl, out)
        sp.n_partials = 1; sp.partials[0].out = pout.data_ptr(); sp.partials[0].lse = plse.data_ptr(); sp.partials[0].count = 1
        sus = timeit(lambda: _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), stream)), a.iters)
        sby = 2 * 2 * Hkv * D * B * a.S + 2 * B * Hq * D * 2
        for P in (1024, 4096, 16384):
            sk = torch.randn(1, P, Hkv, D, device=dev, dtype=dt); sv = torch.randn_like(sk)
            lse = torch.empty(B * Hq, device=dev, dtype=torch.float32)
            p = PrefixParams()
            p.q, p.k, p.v, p.out, p.lse = q.data_ptr(), sk.data_ptr(), sv.data_ptr(), out.data_ptr(), lse.data_ptr()
            p.k_group_stride, p.k_tok_stride, p.k_head_stride = sk.stride(0), sk.stride(1), sk.stride(2)
            p.v_group_stride, p.v_tok_stride, p.v_head_stride = sv.stride(0), sv.stride(1), sv.stride(2)
            p.dtype = _dtype_code(q); p.B, p.nq, p.Hq, p.Hkv, p.D = B, 1, Hq, Hkv, D
            p.sb, p.kv_len, p.lse_layout, p.num_splits = 1, P, HYD_LSE_BQH, 0
            n = lib.hyd_prefix_workspace_bytes(C.byref(p))
            if n:
                ws = torch.empty(n, dtype=torch.uint8, device=dev); p.workspace, p.workspace_bytes = ws.data_ptr(), n
            ns = C.c_int32(); gr = C.c_int32(); sl_ = C.c_int32()
            lib.hyd_prefix_plan(C.byref(p), C.byref(ns), C.byref(gr), C.byref(sl_))
            us = timeit(lambda: _lib.check(lib.hyd_prefix_attn_fwd(C.byref(p), stream)), a.iters)
            fl = 4.0 * B * Hq * P * D
            print(f"{Hq:3d}/{Hkv:<2d} {B:5d} {P:6d} | {us:9.1f} {fl/us/1e6:7.1f} {ns.value:6d} | {sus:17.1f} {sby/sus/1e3:7.0f}")
