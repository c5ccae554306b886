#!/usr/bin/env python3
"""Probe (development, ablation library): does a HIP GRAPH make the two-stream form pay?  The fork / join around
prefix pass (stream A) || suffix pass (stream B) + combine costs ~25 us per step when issued eagerly (overlap_exp.py);
here the same choreography is CAPTURED (the fork and the join become graph edges) and replayed, next to the captured
in-order fused call.

    HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so python tests/probes/overlap_graph.py
"""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

from hydragen_amd import _lib
from hydragen_amd._lib import DecodeParams, PrefixParams, SuffixParams, HYD_LSE_BQH
from hydragen_amd.attention import _fill_level
from hydragen_amd.flash import fill_suffix_params, _dtype_code

lib = _lib.load()
dev = "cuda:0"
B, P, Smax, Hq, Hkv, D = 1024, 2048, 128, 32, 32, 128
dt = torch.bfloat16
torch.manual_seed(0)
q = torch.randn(B, 1, Hq, D, device=dev, dtype=dt)
k = torch.randn(B, Smax, Hkv, D, device=dev, dtype=dt)
v = torch.randn_like(k)
sk = torch.randn(1, P, Hkv, D, device=dev, dtype=dt)
sv = torch.randn_like(sk)
out = torch.empty_like(q)
out_p, out_s = torch.empty_like(q), torch.empty_like(q)
lse_p = torch.empty(B, 1, Hq, device=dev, dtype=torch.float32)
lse_s = torch.empty_like(lse_p)

pp = PrefixParams()
pp.q, pp.k, pp.v, pp.out, pp.lse = q.data_ptr(), sk.data_ptr(), sv.data_ptr(), out_p.data_ptr(), lse_p.data_ptr()
pp.k_group_stride, pp.k_tok_stride, pp.k_head_stride = sk.stride(0), sk.stride(1), sk.stride(2)
pp.v_group_stride, pp.v_tok_stride, pp.v_head_stride = sv.stride(0), sv.stride(1), sv.stride(2)
pp.dtype = _dtype_code(q)
pp.B, pp.nq, pp.Hq, pp.Hkv, pp.D = B, 1, Hq, Hkv, D
pp.sb, pp.kv_len, pp.lse_layout, pp.num_splits = 1, P, HYD_LSE_BQH, 1
op = (C.c_void_p * 2)(out_p.data_ptr(), out_s.data_ptr())
lp = (C.c_void_p * 2)(lse_p.data_ptr(), lse_s.data_ptr())
rows = B * Hq


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def replay_us(g, iters=30):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def overlapped(sp, sa, sb_):
    main = torch.cuda.current_stream()
    sa.wait_stream(main)
    sb_.wait_stream(main)
    _lib.check(lib.hyd_prefix_attn_fwd(C.byref(pp), sa.cuda_stream))
    _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), sb_.cuda_stream))
    main.wait_stream(sa)
    main.wait_stream(sb_)
    _lib.check(lib.hyd_combine_lse(op, lp, 2, rows, D, _dtype_code(q), out.data_ptr(), None, main.cuda_stream))


def overlapped1(sp, sa):
    """suffix stays on the capturing stream; only the prefix pass forks"""
    main = torch.cuda.current_stream()
    sa.wait_stream(main)
    _lib.check(lib.hyd_prefix_attn_fwd(C.byref(pp), sa.cuda_stream))
    _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), main.cuda_stream))
    main.wait_stream(sa)
    _lib.check(lib.hyd_combine_lse(op, lp, 2, rows, D, _dtype_code(q), out.data_ptr(), None, main.cuda_stream))


for s in (16, 64, 128):
    sl = torch.full((B,), s, dtype=torch.int32, device=dev)
    sp = SuffixParams()
    fill_suffix_params(sp, q, k, v, sl, out_s)
    sp.lse = lse_s.data_ptr()
    fp = DecodeParams()
    fill_suffix_params(fp.suffix, q, k, v, sl, out)
    fp.n_levels = 1
    _fill_level(fp.levels[0], sk, sv, None, None, False, B)
    n = lib.hyd_decode_workspace_bytes(C.byref(fp))
    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=dev)
    fp.workspace, fp.workspace_bytes = ws.data_ptr(), n
    os.environ.pop("HYD_PREFIX_PERSIST", None)
    g0 = capture(lambda: _lib.check(lib.hyd_decode_attn_fused(C.byref(fp), torch.cuda.current_stream().cuda_stream)))
    t0 = replay_us(g0)
    want = out.clone()
    line = f"S={s:4d} graph, in order {t0:7.1f} us | graph, forked:"
    for n_p in (0, 48, 64, 96, 128):
        if n_p:
            os.environ["HYD_PREFIX_PERSIST"] = str(n_p)
        else:
            os.environ.pop("HYD_PREFIX_PERSIST", None)
        for form in (2, 1):
            sa, sb_ = torch.cuda.Stream(), torch.cuda.Stream()
            g = capture((lambda: overlapped(sp, sa, sb_)) if form == 2 else (lambda: overlapped1(sp, sa)))
            out.zero_()
            t = replay_us(g)
            err = float((out.float() - want.float()).abs().max())
            line += f"  Np={n_p}/{form}br {t:6.1f}" + ("" if err < 2e-2 else f"(ERR {err:.2g})")
            del g
    print(line, flush=True)
