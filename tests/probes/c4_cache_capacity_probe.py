#!/usr/bin/env python3
"""Probe (GPU box): BASELINE config 4 (two-level hierarchy, B = 1024, suffix 32, 32/32 heads) with the unique K/V cache's CAPACITY
(= the distance between sequences) varied while the 32 valid rows stay the same: exact-size caches (32 rows, 256 KB between
sequences: what tools/bench_configs.py allocates) against 64 / 128 / 256-row caches and the model's [B, K|V, rows] arena."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from hydragen_amd.attention import hydragen_attention_nopad
from hydragen_amd import placement

DEV, dt = "cuda:0", torch.bfloat16
B, S, H, D = 1024, 32, 32, 128
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s: torch.randn(*s, device=DEV, dtype=dt, generator=g)
q = r(B, 1, H, D)
sks, svs = [r(1, 1024, H, D), r(32, 64, H, D)], [r(1, 1024, H, D), r(32, 64, H, D)]
lens = torch.full((B,), S, dtype=torch.int32, device=DEV)


def timed(fn, iters=40):
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


print("| cache | rows between sequences | whole operator us (b2b median) |")
print("|---|---|---|")
for name, cap, arena in (("two tensors", 32, False), ("two tensors", 48, False), ("two tensors", 64, False), ("two tensors", 128, False), ("two tensors", 256, False),
                         ("arena [B, K|V, rows]", 32, True), ("arena [B, K|V, rows]", 64, True), ("arena [B, K|V, rows]", 128, True)):
    if arena:
        a = placement.kv_arena((B, cap, H, D), dt, DEV, zero=False)
        a.normal_()
        k, v = a[0], a[1]
        dist = 2 * cap
    else:
        k, v = r(B, cap, H, D), r(B, cap, H, D)
        dist = cap
    fn = lambda: hydragen_attention_nopad(q, k[:, :S] if cap == S else k, v[:, :S] if cap == S else v, sks, svs, seq_len=lens)
    print(f"| {name}, {cap} rows | {dist} | {timed(fn):7.1f} |", flush=True)
