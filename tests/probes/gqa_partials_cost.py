#!/usr/bin/env python3
"""Probe: what the epilogue's partial merge costs the grouped-query suffix kernel -- no partial, one 16-bit partial
(prefetched under the K/V stream), N fp32 split slices (read in the epilogue).  C5 (B=2048, 8/1, S=256) and C3
(B=64, 32/8, S=256) shapes; back to back and cold (4 KV sets in rotation)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hydragen_amd import _lib
from hydragen_amd._lib import SuffixParams
from hydragen_amd.flash import fill_suffix_params
lib = _lib.load(); dev = "cuda:0"; dt = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream

def timeit(fns, iters=40):
    for i in range(8): fns[i % len(fns)]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (B, Hq, Hkv, S, nsl) in ((2048, 8, 1, 256, 2), (64, 32, 8, 256, 16), (1024, 32, 32, 32, 2)):
    q = torch.randn(B, 1, Hq, 128, device=dev, dtype=dt)
    ks = [torch.randn(B, S, Hkv, 128, device=dev, dtype=dt) for _ in range(4)]
    vs = [torch.randn_like(k) for k in ks]
    out = torch.empty_like(q)
    sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    rows = B * Hq
    p16 = torch.randn_like(q); l16 = torch.randn(B, 1, Hq, device=dev)
    ob = (rows * 128 * 4 + 255) // 256 * 256; lb = (rows * 4 + 255) // 256 * 256
    pf = torch.randn(nsl * ob // 4, device=dev); lf = torch.randn(nsl * lb // 4, device=dev)
    line = f"B={B} {Hq}/{Hkv} S={S}:"
    for name, setup in (("no partial", 0), ("one 16-bit", 1), ("two 16-bit", 3), (f"{nsl} fp32 slices", 2)):
        calls = []
        for k, v in zip(ks, vs):
            sp = SuffixParams(); fill_suffix_params(sp, q, k, v, sl, out)
            if setup == 1 or setup == 3:
                sp.n_partials = 1 if setup == 1 else 2
                for i in range(sp.n_partials):
                    sp.partials[i].out = p16.data_ptr(); sp.partials[i].lse = l16.data_ptr(); sp.partials[i].count = 1
            elif setup == 2:
                sp.n_partials = 1
                sp.partials[0].out = pf.data_ptr(); sp.partials[0].lse = lf.data_ptr(); sp.partials[0].count = nsl; sp.partials[0].is_f32 = 1
            calls.append(lambda sp=sp: _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), stream)))
        line += f"  {name}: {timeit(calls[:1]):6.1f} hot / {timeit(calls):6.1f} cold"
    print(line, flush=True)
