#!/usr/bin/env python3
"""Probe (GPU box): the suffix pass's HBM rate by head dim and head configuration (B = 1024, suffix 64 and 128 in 128-row caches
of the model's arena layout): which kernel each shape takes is the library's own choice."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from hydragen_amd import placement
from hydragen_amd.flash import flash_attention_seqlen

DEV, dt = "cuda:0", torch.bfloat16
B, cap = 1024, 128


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    evs[0].record()
    for i in range(iters):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    return torch.tensor([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(iters)]).median().item()


print("| head dim | q / kv heads | S | us | TB/s (K/V + q + out) |")
print("|---|---|---|---|---|")
for D, Hq, Hkv in ((64, 32, 32), (64, 32, 8), (64, 64, 8), (128, 32, 32), (128, 32, 8), (128, 64, 8), (256, 16, 16), (256, 16, 4), (256, 32, 8), (96, 32, 32), (80, 32, 8)):
    g = torch.Generator(device=DEV).manual_seed(0)
    q = torch.randn(B, 1, Hq, D, device=DEV, dtype=dt, generator=g)
    a = placement.kv_arena((B, cap, Hkv, D), dt, DEV, zero=False)
    a.normal_()
    for S in (64, 128):
        lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
        us = timed(lambda: flash_attention_seqlen(q, a[0], a[1], lens))
        byts = B * S * Hkv * D * 2 * 2 + 2 * B * Hq * D * 2
        print(f"| {D} | {Hq} / {Hkv} | {S} | {us:7.1f} | {byts / us / 1e6:5.2f} |", flush=True)
    del a
    torch.cuda.empty_cache()
