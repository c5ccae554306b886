#!/usr/bin/env python3
"""Development tool (ablation library): the persistent matrix-core suffix kernel (suffix_gqa_stream.h) against the
shipped suffix kernels -- same call (hyd_suffix_attn_fwd with one 16-bit prefix partial), outputs compared, both timed
back to back and cold (4 KV sets in rotation).   HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so python tools/gqa_stream_bench.py"""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hydragen_amd import _lib
from hydragen_amd._lib import SuffixParams
from hydragen_amd.flash import fill_suffix_params
lib = _lib.load(); dev = "cuda:0"; dt = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream

def timeit(fns, iters=24):
    for i in range(4): fns[i % len(fns)]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def setenv(**kw):
    for k_, v_ in kw.items():
        if v_ is None: os.environ.pop(k_, None)
        else: os.environ[k_] = str(v_)

shapes = [(1024, 32, 32, 16, False), (1024, 32, 32, 64, False), (1024, 32, 32, 128, False), (2048, 8, 1, 256, False),
          (2048, 8, 1, 64, False), (1024, 32, 8, 128, False), (1024, 32, 32, 100, True), (2048, 8, 1, 200, True)]
if len(sys.argv) > 1 and sys.argv[1] == "quick": shapes = shapes[2:4]
for (B, Hq, Hkv, S, ragged) in shapes:
    NSET = 4 if B * S * Hkv * 128 * 4 > 64e6 else 1
    q = torch.randn(B, 1, Hq, 128, device=dev, dtype=dt)
    ks = [torch.randn(B, S, Hkv, 128, device=dev, dtype=dt) for _ in range(NSET)]
    vs = [torch.randn_like(k) for k in ks]
    out = torch.empty_like(q); lse = torch.empty(B, 1, Hq, device=dev, dtype=torch.float32)
    pout = torch.randn_like(q); plse = torch.randn(B, 1, Hq, device=dev, dtype=torch.float32)
    if ragged:
        g = torch.Generator().manual_seed(S); sl = torch.randint(0, S + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    else:
        sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    sps = []
    for k, v in zip(ks, vs):
        sp = SuffixParams(); fill_suffix_params(sp, q, k, v, sl, out); sp.lse = lse.data_ptr()
        sp.n_partials = 1; sp.partials[0].out = pout.data_ptr(); sp.partials[0].lse = plse.data_ptr(); sp.partials[0].count = 1
        sps.append(sp)
    calls = [(lambda sp=sp: _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), stream))) for sp in sps]
    by = float(sl.sum()) * 2 * 2 * Hkv * 128 + 3 * B * Hq * 128 * 2
    setenv(HYD_STREAM_WGS=None, HYD_STREAM_NBUF=None)
    t_hot, t_cold = timeit(calls[:1]), timeit(calls)
    calls[0](); torch.cuda.synchronize(); want, wl = out.clone(), lse.clone()
    line = f"B={B} {Hq}/{Hkv} S={S}{' ragged' if ragged else ''}: shipped {t_hot:6.1f} hot {t_cold:6.1f} cold ({by / t_cold / 1e6:4.2f} TB/s) | persistent:"
    for wgs in (1024, 1536, 2048):
        for upi in (4, 16):
            setenv(HYD_STREAM_WGS=wgs, HYD_STREAM_NBUF=100, HYD_CORUN_UPI=upi)
            out.zero_(); lse.zero_()
            th, tc = timeit(calls[:1]), timeit(calls)
            calls[0](); torch.cuda.synchronize()
            err = float((out.float() - want.float()).abs().max()); lerr = float((lse - wl).abs().nan_to_num(0.0, 0.0, 0.0).max())
            line += f" w{wgs}/u{upi} {th:6.1f}/{tc:6.1f}" + ("" if err < 1e-2 and lerr < 1e-3 else f"(ERR {err:.2g} {lerr:.2g})")
    setenv(HYD_STREAM_WGS=None, HYD_STREAM_NBUF=None)
    print(line, flush=True)
