#!/usr/bin/env python3
"""Probe (GPU box): the suffix pass at C2 heads (B = 1024, 32/32, 128-row caches in the model's arena layout) with RAGGED lengths
against uniform ones of the same total: uniform 64 | uniform random 1..128 (mean ~64) | half 16 / half 112 | sorted ascending / descending."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from hydragen_amd import placement
from hydragen_amd.flash import flash_attention_seqlen, longest_first, seq_order

DEV, dt = "cuda:0", torch.bfloat16
# python ragged_lengths_probe.py [B Hq Hkv cap]   (default: C2 heads, 128-row caches; "2048 64 8 256" = whole-job C5)
B, HQ, H, cap = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 32, 32, 128)
D = 128
g = torch.Generator(device=DEV).manual_seed(0)
q = torch.randn(B, 1, HQ, D, device=DEV, dtype=dt, generator=g)
(arena,), _ = placement.place_kv_arenas(1, (B, cap, H, D), dt, DEV, HQ, zero=False)
arena.normal_()


def timed(lens, iters=40):
    fn = lambda: flash_attention_seqlen(q, arena[0], arena[1], lens)
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


rnd = torch.randint(1, cap + 1, (B,), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
lo, hi = cap // 8, cap - cap // 8
cases = {
    f"uniform {cap // 2}": torch.full((B,), cap // 2, dtype=torch.int32),
    f"random 1..{cap}": rnd,
    "random, sorted ascending": rnd.sort().values,
    "random, sorted descending": rnd.sort(descending=True).values,
    f"half {lo} / half {hi} (interleaved)": torch.tensor([lo, hi] * (B // 2), dtype=torch.int32),
    f"half {lo} then half {hi}": torch.tensor([lo] * (B // 2) + [hi] * (B // 2), dtype=torch.int32),
}
print("| lengths | keys in total | us (b2b median) | TB/s of K/V + q / out | us with flash.seq_order(longest_first(lens)) | us with the identity order |")
print("|---|---|---|---|---|---|")
ident = torch.arange(B, dtype=torch.int32, device=DEV)
for name, lens in cases.items():
    tot = int(lens.sum())
    ld = lens.to(DEV)
    us = timed(ld)
    with seq_order(longest_first(ld)):
        us_o = timed(ld)
    with seq_order(ident):
        us_i = timed(ld)
    byts = tot * H * D * 2 * 2 + 2 * B * HQ * D * 2
    print(f"| {name} | {tot} | {us:7.1f} | {byts / us / 1e6:5.2f} | {us_o:7.1f} | {us_i:7.1f} |", flush=True)
