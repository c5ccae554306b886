#!/usr/bin/env python3
"""Development tool (GPU box): the rate of the suffix pass over successive K|V arenas of one process -- the map behind
hydragen_amd/placement.py.  Allocates `--count` arenas (all alive, so each is new memory), times the library's own suffix
pass on each at half and at all of the rows, prints the series.

    python tools/placement_map.py --shape c5 --count 120"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hydragen_amd.flash import flash_attention_seqlen

SHAPES = {"c2": (1024, 128, 32, 32), "c5": (2048, 256, 64, 8)}  # batch, cache rows, q heads, kv heads


def t_us(q, k, v, lens, iters=3):
    best = 1e30
    for i in range(iters + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flash_attention_seqlen(q, k, v, lens)
        e1.record()
        e1.synchronize()
        if i:
            best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="c5", choices=list(SHAPES))
    ap.add_argument("--count", type=int, default=100)
    a = ap.parse_args()
    B, S, Hq, Hkv = SHAPES[a.shape]
    dev, D = "cuda:0", 128
    q = torch.zeros(B, 1, Hq, D, device=dev, dtype=torch.bfloat16)
    half = torch.full((B,), S // 2, dtype=torch.int32, device=dev)
    full = torch.full((B,), S, dtype=torch.int32, device=dev)
    arenas, rows = [], []
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(a.count):
        if torch.cuda.mem_get_info()[0] < 6 << 30:
            break
        kv = torch.empty(2, B, S, Hkv, D, device=dev, dtype=torch.bfloat16)
        arenas.append(kv)
        rows.append((t_us(q, kv[0], kv[1], half), t_us(q, kv[0], kv[1], full)))
    gib = arenas[0].numel() * 2 / 2**30
    print(f"# {a.shape}: {len(arenas)} arenas of {gib:.1f} GiB (free at start {free0 / 2**30:.0f} GiB); us at half / all rows")
    for i in range(0, len(rows), 8):
        print(f"#{i:3d} " + "  ".join(f"{h:6.1f}/{f:6.1f}" for h, f in rows[i:i + 8]))
    # second pass over the same arenas: is the map stable?
    again = [t_us(q, kv[0], kv[1], half) for kv in arenas]
    drift = max(abs(x - r[0]) for x, r in zip(again, rows))
    hs = sorted(r[0] for r in rows)
    print(f"# half-rows: min {hs[0]:.1f} median {hs[len(hs) // 2]:.1f} max {hs[-1]:.1f}; second pass differs by at most {drift:.1f} us")


if __name__ == "__main__":
    main()
