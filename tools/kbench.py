#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box (development tool):
    python tools/kbench.py prefix [--P 256,512,...] [--B 1024 --Hq 32 --Hkv 32]
    python tools/kbench.py suffix [--S 1,2,4,...]
Prints mean us per launch (HIP events around `iters` back-to-back launches)."""
import argparse, sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hydragen_amd import _lib
from hydragen_amd._lib import PrefixParams, SuffixParams, HYD_LSE_BQH
from hydragen_amd.flash import fill_suffix_params, _dtype_code

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["prefix", "suffix", "fused"])
ap.add_argument("--P", default="128,256,512,1024,2048,4096,8192")
ap.add_argument("--S", default="1,2,4,8,16,32,64,128,256")
ap.add_argument("--B", type=int, default=1024)
ap.add_argument("--Hq", type=int, default=32)
ap.add_argument("--Hkv", type=int, default=32)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--splits", type=int, default=0, help="0 = the library's own split-KV plan")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--slices", type=int, default=0, help="fused: hand the suffix pass N stacked fp32 slices (a split prefix level) instead of one 16-bit partial")
ap.add_argument("--partials", type=int, default=1, help="fused: number of separate 16-bit partials (shared levels) handed to the suffix pass")
ap.add_argument("--kv-gap", type=int, default=0, help="suffix: bytes between the end of K and the start of V in their arena (-1: two allocations)")
ap.add_argument("--smax", type=int, default=0, help="suffix: token rows of the cache allocation (default: the largest --S)")
ap.add_argument("--pad-tokens", type=int, default=0, help="suffix: extra token rows per sequence in the cache allocation (batch stride (S + pad) rows)")
a = ap.parse_args()
lib = _lib.load()
dev = "cuda:0"
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
stream = torch.cuda.current_stream().cuda_stream

def timeit(fn, iters):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

q = torch.randn(a.B, 1, a.Hq, a.D, device=dev, dtype=dt)
if a.what == "prefix":
    for P in map(int, a.P.split(",")):
        sk = torch.randn(1, P, a.Hkv, a.D, device=dev, dtype=dt); sv = torch.randn_like(sk)
        out = torch.empty_like(q); lse = torch.zeros(a.B * a.Hq + 64, device=dev, dtype=torch.float32)
        p = PrefixParams()
        p.q, p.k, p.v, p.out, p.lse = q.data_ptr(), sk.data_ptr(), sv.data_ptr(), out.data_ptr(), lse.data_ptr()
        p.k_group_stride, p.k_tok_stride, p.k_head_stride = sk.stride(0), sk.stride(1), sk.stride(2)
        p.v_group_stride, p.v_tok_stride, p.v_head_stride = sv.stride(0), sv.stride(1), sv.stride(2)
        p.dtype = _dtype_code(q); p.B, p.nq, p.Hq, p.Hkv, p.D = a.B, 1, a.Hq, a.Hkv, a.D
        p.sb, p.kv_len, p.lse_layout, p.num_splits = 1, P, HYD_LSE_BQH, a.splits
        n = lib.hyd_prefix_workspace_bytes(C.byref(p))
        if n:
            ws = torch.empty(n, dtype=torch.uint8, device=dev); p.workspace, p.workspace_bytes = ws.data_ptr(), n
        us = timeit(lambda: _lib.check(lib.hyd_prefix_attn_fwd(C.byref(p), stream)), a.iters)
        fl = 4.0 * a.B * a.Hq * P * a.D
        print(f"prefix P={P:6d}  {us:9.2f} us  {fl/us/1e6:8.1f} TFLOP/s  ({fl/us/1e6/2500*100:5.1f}% of 2.5PF)")
        import os
        if int(os.environ.get("HYD_DBG", "0")) & 2048:  # ablation build: cycle stamps of the 4 waves of workgroup 0
            torch.cuda.synchronize()
            ts = lse.view(torch.int32).flatten()[a.B * a.Hq:a.B * a.Hq + 64].cpu().view(8, 8)
            for w in range(4):
                d = [(int(ts[w, k]) - int(ts[w, 0])) & 0xffffffff for k in range(6)]
                print(f"   wave {w}: setup+issue {d[1]}  first wait+barrier {d[2]-d[1]}  loop {d[3]-d[2]}  merge write+barrier {d[4]-d[3]}  combine+stores {d[5]-d[4]}  | kernel {d[5]} ticks")
else:
    Smax = max(max(map(int, a.S.split(","))), a.smax)
    n_ = a.B * (Smax + a.pad_tokens) * a.Hkv * a.D
    if a.kv_gap < 0:  # two separate allocations
        k = torch.randn(a.B, Smax + a.pad_tokens, a.Hkv, a.D, device=dev, dtype=dt)[:, :Smax]; v = torch.randn(a.B, Smax + a.pad_tokens, a.Hkv, a.D, device=dev, dtype=dt)[:, :Smax]
    else:  # one arena, K | gap | V (PerLayerKVCache allocates K | V)
        arena = torch.randn(2 * n_ + a.kv_gap // 2, device=dev, dtype=dt)
        k = arena[:n_].view(a.B, Smax + a.pad_tokens, a.Hkv, a.D)[:, :Smax]; v = arena[n_ + a.kv_gap // 2:].view(a.B, Smax + a.pad_tokens, a.Hkv, a.D)[:, :Smax]
    out = torch.empty_like(q); lse = torch.empty(a.B, 1, a.Hq, device=dev, dtype=torch.float32)
    pout = torch.randn_like(q); plse = torch.randn(a.B, 1, a.Hq, device=dev, dtype=torch.float32)
    for S in map(int, a.S.split(",")):
        sl = torch.full((a.B,), S, dtype=torch.int32, device=dev)
        sp = SuffixParams(); fill_suffix_params(sp, q, k, v, sl, out)
        if a.what == "fused" and a.slices > 0:
            rows_ = a.B * a.Hq; al_ = lambda x: (x + 255) // 256 * 256
            if S == int(a.S.split(",")[0]):
                sl_o = torch.randn(a.slices * al_(rows_ * a.D * 4) // 4, device=dev, dtype=torch.float32); sl_l = torch.randn(a.slices * al_(rows_ * 4) // 4, device=dev, dtype=torch.float32)
            sp.n_partials = 1; sp.partials[0].out = sl_o.data_ptr(); sp.partials[0].lse = sl_l.data_ptr(); sp.partials[0].count = a.slices; sp.partials[0].is_f32 = 1
        elif a.what == "fused":
            if S == int(a.S.split(",")[0]):
                pouts = [torch.randn_like(q) for _ in range(a.partials)]; plses = [torch.randn(a.B, 1, a.Hq, device=dev, dtype=torch.float32) for _ in range(a.partials)]
            sp.n_partials = a.partials
            for i_ in range(a.partials):
                sp.partials[i_].out = pouts[i_].data_ptr(); sp.partials[i_].lse = plses[i_].data_ptr(); sp.partials[i_].count = 1; sp.partials[i_].is_f32 = 0
        else:
            sp.lse = lse.data_ptr()
        us = timeit(lambda: _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), stream)), a.iters)
        by = 2 * 2 * a.Hkv * a.D * a.B * S + 2 * a.B * a.Hq * a.D * 2 + 4 * a.B * a.Hq
        print(f"{a.what} S={S:5d}  {us:9.2f} us  {by/us/1e3:8.1f} GB/s  ({by/us/1e3/8000*100:5.1f}% of 8TB/s)")
