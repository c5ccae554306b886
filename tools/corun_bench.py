#!/usr/bin/env python3
"""Development tool (ablation library): the co-run kernel against the in-order form, and its streaming role alone.

    HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so python tools/corun_bench.py [stream] [corun]
"""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hydragen_amd import _lib
from hydragen_amd._lib import DecodeParams, SuffixParams
from hydragen_amd.attention import _fill_level
from hydragen_amd.flash import fill_suffix_params

lib = _lib.load()
dev = "cuda:0"
B, P, Smax, H, D = 1024, 2048, 128, 32, 128
dt = torch.bfloat16
torch.manual_seed(0)
q = torch.randn(B, 1, H, D, device=dev, dtype=dt)
k = torch.randn(B, Smax, H, D, device=dev, dtype=dt)
v = torch.randn_like(k)
sk = torch.randn(1, P, H, D, device=dev, dtype=dt)
sv = torch.randn_like(sk)
stream = torch.cuda.current_stream().cuda_stream
what = sys.argv[1:] or ["stream", "corun"]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def setenv(**kw):
    for k_, v_ in kw.items():
        if v_ is None:
            os.environ.pop(k_, None)
        else:
            os.environ[k_] = str(v_)


def lens_for(s, ragged):
    if not ragged:
        return torch.full((B,), s, dtype=torch.int32, device=dev)
    g = torch.Generator(device="cpu").manual_seed(s)
    return torch.randint(0, s + 1, (B,), generator=g, dtype=torch.int32).to(dev)


if "stream" in what:
    out = torch.empty_like(q)
    lse = torch.empty(B, 1, H, device=dev, dtype=torch.float32)
    for s, ragged in ((16, False), (64, False), (128, False), (100, True)):
        sl = lens_for(s, ragged)
        sp = SuffixParams()
        fill_suffix_params(sp, q, k, v, sl, out)
        sp.lse = lse.data_ptr()
        call = lambda: _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), stream))
        setenv(HYD_STREAM_WGS=None)
        t0 = timeit(call)
        want, wl = out.clone(), lse.clone()
        by = float(sl.sum()) * 2 * 2 * H * D + 2 * B * H * D * 2
        line = f"suffix S={s}{' ragged' if ragged else ''}: occupancy kernel {t0:7.1f} us ({by / t0 / 1e6:5.2f} TB/s) | streaming role:"
        for nbuf in (2, 4, 8):
            for wgs in (160, 256):
                setenv(HYD_STREAM_WGS=wgs, HYD_STREAM_NBUF=nbuf)
                out.zero_()
                t = timeit(call)
                err = float((out.float() - want.float()).abs().max())
                lerr = float((lse - wl).abs().nan_to_num(0.0, 0.0, 0.0).max())
                line += f" nbuf{nbuf}/wg{wgs} {t:6.1f}" + ("" if err < 1e-2 and lerr < 1e-3 else f"(ERR {err:.2g} {lerr:.2g})")
        setenv(HYD_STREAM_WGS=None)
        print(line, flush=True)

if "corun" in what:
    out = torch.empty_like(q)
    for s, ragged in ((4, False), (16, False), (32, False), (64, False), (96, False), (128, False), (100, True)):
        sl = lens_for(s, ragged)
        p = DecodeParams()
        fill_suffix_params(p.suffix, q, k, v, sl, out)
        p.n_levels = 1
        _fill_level(p.levels[0], sk, sv, None, None, False, B)
        n = lib.hyd_decode_workspace_bytes(C.byref(p))
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        p.workspace, p.workspace_bytes = ws.data_ptr(), n
        call = lambda: _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))
        setenv(HYD_CORUN=None)
        t0 = timeit(call)
        want = out.clone()
        line = f"decode S={s}{' ragged' if ragged else ''}: in order {t0:7.1f} us | co-run:"
        for upi in (8, 16):
            for np_ in (1, 2, 3, 4, 6):
                setenv(HYD_CORUN=1, HYD_CORUN_NP=np_, HYD_CORUN_UPI=upi)
                out.zero_()
                t = timeit(call)
                err = float((out.float() - want.float()).abs().max())
                line += f" upi{upi}/np{np_} {t:6.1f}" + ("" if err < 1.6e-2 else f"(ERR {err:.2g})")
        setenv(HYD_CORUN=None)
        print(line, flush=True)
