#!/usr/bin/env python3
"""Probe (development, ablation library): the shared phase as PERSISTENT prefix workgroups on a part of the chip
(stream A, launched first) while the unique phase streams on the rest (stream B); partials merged by a combine
launch after the join.  Compared with the in-order fused call.

    HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so python tests/probes/overlap_exp.py
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
if len(sys.argv) > 1:
    B, P, Smax, Hq, Hkv = map(int, sys.argv[1:6])
dt = torch.bfloat16
torch.manual_seed(0)
q = torch.randn(B, 1, Hq, D, device=dev, dtype=dt)
k = torch.randn(B, Smax, Hkv, D, device=dev, dtype=dt)
v = torch.randn_like(k)
sk = torch.randn(1, P, Hkv, D, device=dev, dtype=dt)
sv = torch.randn_like(sk)
out = torch.empty_like(q)
out_p = torch.empty_like(q)
lse_p = torch.empty(B, 1, Hq, device=dev, dtype=torch.float32)
out_s = torch.empty_like(q)
lse_s = torch.empty(B, 1, Hq, device=dev, dtype=torch.float32)
main = torch.cuda.current_stream()
sa, sb_ = torch.cuda.Stream(), torch.cuda.Stream()

pp = PrefixParams()
pp.q, pp.k, pp.v, pp.out, pp.lse = q.data_ptr(), sk.data_ptr(), sv.data_ptr(), out_p.data_ptr(), lse_p.data_ptr()
pp.k_group_stride, pp.k_tok_stride, pp.k_head_stride = sk.stride(0), sk.stride(1), sk.stride(2)
pp.v_group_stride, pp.v_tok_stride, pp.v_head_stride = sv.stride(0), sv.stride(1), sv.stride(2)
pp.dtype = _dtype_code(q)
pp.B, pp.nq, pp.Hq, pp.Hkv, pp.D = B, 1, Hq, Hkv, D
pp.sb, pp.kv_len, pp.lse_layout, pp.num_splits = 1, P, HYD_LSE_BQH, 1


def mk_suffix(s):
    sl = torch.full((B,), s, dtype=torch.int32, device=dev)
    sp = SuffixParams()
    fill_suffix_params(sp, q, k, v, sl, out_s)
    sp.lse = lse_s.data_ptr()
    return sp, sl


def mk_fused(s):
    sl = torch.full((B,), s, dtype=torch.int32, device=dev)
    p = DecodeParams()
    fill_suffix_params(p.suffix, q, k, v, sl, out)
    p.n_levels = 1
    _fill_level(p.levels[0], sk, sv, None, None, False, B)
    n = lib.hyd_decode_workspace_bytes(C.byref(p))
    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), n
    return p, sl, ws


op = (C.c_void_p * 2)(out_p.data_ptr(), out_s.data_ptr())
lp = (C.c_void_p * 2)(lse_p.data_ptr(), lse_s.data_ptr())
rows = B * Hq


def timeit(fn, iters=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def prefix_on(stream):
    _lib.check(lib.hyd_prefix_attn_fwd(C.byref(pp), stream.cuda_stream))


evf = torch.cuda.Event()
eva, evb = torch.cuda.Event(), torch.cuda.Event()


def overlapped(sp, order="ps"):
    evf.record(main)
    sa.wait_event(evf)
    sb_.wait_event(evf)
    if order == "ps":
        prefix_on(sa)
        _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), sb_.cuda_stream))
    else:
        _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), sb_.cuda_stream))
        prefix_on(sa)
    eva.record(sa)
    evb.record(sb_)
    main.wait_event(eva)
    main.wait_event(evb)
    _lib.check(lib.hyd_combine_lse(op, lp, 2, rows, D, _dtype_code(q), out.data_ptr(), None, main.cuda_stream))


def serial3(sp):
    prefix_on(main)
    _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), main.cuda_stream))
    _lib.check(lib.hyd_combine_lse(op, lp, 2, rows, D, _dtype_code(q), out.data_ptr(), None, main.cuda_stream))


print(f"B={B} P={P} Hq={Hq} Hkv={Hkv} D={D}")
os.environ.pop("HYD_PREFIX_PERSIST", None)
t_pre = timeit(lambda: prefix_on(main))
print(f"prefix alone, one workgroup per unit: {t_pre:.1f} us")
for n_p in (32, 64, 96, 128, 192):
    os.environ["HYD_PREFIX_PERSIST"] = str(n_p)
    print(f"prefix alone, {n_p} persistent workgroups: {timeit(lambda: prefix_on(main)):.1f} us")
os.environ.pop("HYD_PREFIX_PERSIST", None)
t_cmb = timeit(lambda: _lib.check(lib.hyd_combine_lse(op, lp, 2, rows, D, _dtype_code(q), out.data_ptr(), None, main.cuda_stream)))
print(f"combine alone: {t_cmb:.1f} us")

for s in (8, 16, 32, 64, 96, 128):
    if s > Smax:
        continue
    sp, sl = mk_suffix(s)
    fp, sl2, ws = mk_fused(s)
    os.environ.pop("HYD_PREFIX_PERSIST", None)
    t_f = timeit(lambda: _lib.check(lib.hyd_decode_attn_fused(C.byref(fp), main.cuda_stream)))
    want = out.clone()
    t_s = timeit(lambda: _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), main.cuda_stream)))
    t_3 = timeit(lambda: serial3(sp))
    line = f"S={s:4d} fused in order {t_f:7.1f} | suffix alone {t_s:7.1f} | 3 launches in order {t_3:7.1f} | overlapped:"
    for n_p in (0, 32, 64, 96, 128, 192):
        if n_p:
            os.environ["HYD_PREFIX_PERSIST"] = str(n_p)
        else:
            os.environ.pop("HYD_PREFIX_PERSIST", None)
        t_o = timeit(lambda: overlapped(sp))
        err = float((out.float() - want.float()).abs().max())
        line += f"  Np={n_p}: {t_o:6.1f}" + ("" if err < 2e-2 else f" (ERR {err:.3g})")
    os.environ["HYD_PREFIX_PERSIST"] = "64"
    t_sp = timeit(lambda: overlapped(sp, "sp"))
    line += f"  | suffix launched first, Np=64: {t_sp:6.1f}"
    print(line, flush=True)
