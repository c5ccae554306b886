#!/usr/bin/env python3
"""Probe: start / end of the persistent prefix pass (stream A) and the suffix pass (stream B) relative to a common
event, one shot per repetition.   HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so python tests/probes/overlap_timeline.py"""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

from hydragen_amd import _lib
from hydragen_amd._lib import PrefixParams, SuffixParams, HYD_LSE_BQH
from hydragen_amd.flash import fill_suffix_params, _dtype_code

lib = _lib.load()
dev = "cuda:0"
B, P, Smax, Hq, Hkv, D = 1024, 2048, 128, 32, 32, 128
dt = torch.bfloat16
q = torch.randn(B, 1, Hq, D, device=dev, dtype=dt)
k = torch.randn(B, Smax, Hkv, D, device=dev, dtype=dt)
v = torch.randn_like(k)
sk = torch.randn(1, P, Hkv, D, device=dev, dtype=dt)
sv = torch.randn_like(sk)
out_p = torch.empty_like(q)
lse_p = torch.empty(B, 1, Hq, device=dev, dtype=torch.float32)
out_s = torch.empty_like(q)
lse_s = torch.empty(B, 1, Hq, device=dev, dtype=torch.float32)
main = torch.cuda.current_stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "prio":
    sa, sb_ = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
else:
    sa, sb_ = torch.cuda.Stream(), torch.cuda.Stream()

pp = PrefixParams()
pp.q, pp.k, pp.v, pp.out, pp.lse = q.data_ptr(), sk.data_ptr(), sv.data_ptr(), out_p.data_ptr(), lse_p.data_ptr()
pp.k_group_stride, pp.k_tok_stride, pp.k_head_stride = sk.stride(0), sk.stride(1), sk.stride(2)
pp.v_group_stride, pp.v_tok_stride, pp.v_head_stride = sv.stride(0), sv.stride(1), sv.stride(2)
pp.dtype = _dtype_code(q)
pp.B, pp.nq, pp.Hq, pp.Hkv, pp.D = B, 1, Hq, Hkv, D
pp.sb, pp.kv_len, pp.lse_layout, pp.num_splits = 1, P, HYD_LSE_BQH, 1
E = lambda: torch.cuda.Event(enable_timing=True)

for s in (64, 128):
    sl = torch.full((B,), s, dtype=torch.int32, device=dev)
    sp = SuffixParams()
    fill_suffix_params(sp, q, k, v, sl, out_s)
    sp.lse = lse_s.data_ptr()
    for n_p in (64, 128):
        os.environ["HYD_PREFIX_PERSIST"] = str(n_p)
        for with_ev in (True, False):
            rows = []
            for rep in range(8):
                e0, xs, xe, ys, ye, ee = E(), E(), E(), E(), E(), E()
                e0.record(main)
                sa.wait_event(e0)
                sb_.wait_event(e0)
                if with_ev:
                    xs.record(sa)
                _lib.check(lib.hyd_prefix_attn_fwd(C.byref(pp), sa.cuda_stream))
                xe.record(sa)
                if with_ev:
                    ys.record(sb_)
                _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), sb_.cuda_stream))
                ye.record(sb_)
                main.wait_event(xe)
                main.wait_event(ye)
                ee.record(main)
                torch.cuda.synchronize()
                if with_ev:
                    rows.append([e0.elapsed_time(t) * 1e3 for t in (xs, xe, ys, ye, ee)])
                else:
                    rows.append([0.0, e0.elapsed_time(xe) * 1e3, 0.0, e0.elapsed_time(ye) * 1e3, e0.elapsed_time(ee) * 1e3])
            r = torch.tensor(rows[3:]).mean(0)
            print(f"{mode} S={s} Np={n_p} start-events={with_ev}: prefix {r[0]:6.1f} -> {r[1]:6.1f} | suffix {r[2]:6.1f} -> {r[3]:6.1f} | joined {r[4]:6.1f} us")
