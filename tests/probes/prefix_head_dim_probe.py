#!/usr/bin/env python3
"""Probe (GPU box): the prefix pass alone (one shared level, no unique keys: attention.py:273-274's early exit) by head dim and head
configuration at B = 1024, P = 2048 -- TFLOP/s of 4 B Hq P D."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from hydragen_amd.attention import hydragen_attention_nopad

DEV, dt = "cuda:0", torch.bfloat16
B, P = 1024, 2048


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


print("| head dim | q / kv heads | us | TFLOP/s | of 2.5 PF |")
print("|---|---|---|---|---|")
for D, Hq, Hkv in ((64, 32, 32), (64, 32, 8), (64, 64, 8), (128, 32, 32), (128, 32, 8), (128, 64, 8), (256, 16, 16), (256, 16, 4), (256, 32, 8)):
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, dtype=dt, generator=g)
    q, sk, sv = r(B, 1, Hq, D), r(1, P, Hkv, D), r(1, P, Hkv, D)
    k = torch.empty(B, 0, Hkv, D, device=DEV, dtype=dt)
    us = timed(lambda: hydragen_attention_nopad(q, k, k, [sk], [sv]))
    fl = 4.0 * B * Hq * P * D
    print(f"| {D} | {Hq} / {Hkv} | {us:7.1f} | {fl / us / 1e6:7.1f} | {fl / us / 1e6 / 2500:5.3f} |", flush=True)
