#!/usr/bin/env python3
"""Probe: would fusing the decode step's projections (q|k|v, gate|up) into one GEMM each pay on MI355X?  M = batch."""
import torch, torch.nn.functional as F
dev, dt = "cuda:0", torch.bfloat16
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, H, I, KVH in ((1024, 4096, 11008, 4096), (2048, 8192, 28672 // 8, 1024 // 8 * 1 + 0)):
    x = torch.randn(M, H, device=dev, dtype=dt)
    nq, nkv = (H, KVH) if M == 1024 else (H // 8, 128)
    wq, wk, wv = (torch.randn(n, H, device=dev, dtype=dt) for n in (nq, nkv, nkv))
    wqkv = torch.cat([wq, wk, wv])
    wg, wu = torch.randn(I, H, device=dev, dtype=dt), torch.randn(I, H, device=dev, dtype=dt)
    wgu = torch.cat([wg, wu])
    a = t(lambda: (F.linear(x, wq), F.linear(x, wk), F.linear(x, wv)))
    b = t(lambda: F.linear(x, wqkv))
    c = t(lambda: (F.linear(x, wg), F.linear(x, wu)))
    d = t(lambda: F.linear(x, wgu))
    print(f"M={M} H={H}: q,k,v separately {a:.1f} us, fused {b:.1f} us | gate,up separately {c:.1f} us, fused {d:.1f} us (I={I})")
