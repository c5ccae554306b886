#!/usr/bin/env python3
"""Host-side cost of one operator call (development tool): wall time per call of a tiny problem, where the
GPU work is negligible, and a cProfile breakdown."""
import cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hydragen_amd.attention import hydragen_attention_nopad
from hydragen_amd.flash import flash_attention_seqlen

dev = "cuda:0"
B, P, S, H, D = 4, 64, 8, 4, 64
r = lambda *s: torch.randn(*s, device=dev, dtype=torch.float16)
q, k, v, sk, sv = r(B, 1, H, D), r(B, S, H, D), r(B, S, H, D), r(1, P, H, D), r(1, P, H, D)
lens = torch.full((B,), S, dtype=torch.int32, device=dev)
f = lambda: hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens)
g = lambda: flash_attention_seqlen(q, k, v, seq_len=lens)
for name, fn in (("hydragen_attention_nopad", f), ("flash_attention_seqlen", g)):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3000 * 1e6:.1f} us per call (host-bound)")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): f()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
