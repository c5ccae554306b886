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
B, H, D, cap = 1024, 32, 128, 128
g = torch.Generator(device=DEV).manual_seed(0)
q = torch.randn(B, 1, H, D, device=DEV, dtype=dt, generator=g)
(arena,), _ = placement.place_kv_arenas(1, (B, cap, H, D), dt, DEV, H, zero=False)
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


rnd = torch.randint(1, 129, (B,), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
cases = {
    "uniform 64": torch.full((B,), 64, dtype=torch.int32),
    "random 1..128": rnd,
    "random, sorted ascending": rnd.sort().values,
    "random, sorted descending": rnd.sort(descending=True).values,
    "half 16 / half 112 (interleaved)": torch.tensor([16, 112] * (B // 2), dtype=torch.int32),
    "half 16 then half 112": torch.tensor([16] * (B // 2) + [112] * (B // 2), dtype=torch.int32),
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
    byts = tot * H * D * 2 * 2 + 2 * B * H * D * 2
    print(f"| {name} | {tot} | {us:7.1f} | {byts / us / 1e6:5.2f} | {us_o:7.1f} | {us_i:7.1f} |", flush=True)
