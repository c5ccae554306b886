#!/usr/bin/env python3
"""The reference's attention microbenchmark sweep (docs/sweeps_from_paper.md:152-170, Figures 5 and 8 of the paper;
scripts/microbenchmark.py): (batch, prefix) in {(512,1024), (1024,2048), (2048,4096), (4096,8192)}, unique suffix
0..512 in steps of 16, the script's default heads (8 query / 1 kv, D=128, scripts/microbenchmark.py:136-138), bf16,
Hydragen vs the no-sharing baseline (private [P+S] KV per sequence).  Protocol of hydragen/benchmark_utils.py:82-170:
the operator is captured into a (HIP) graph, warm-up, per-iteration events, 512 MB cache flush between iterations.

    python tools/paper_microbench.py --out gpurun_out/paper_microbench.md [--step 64]"""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hydragen_amd.attention import hydragen_attention_nopad
from hydragen_amd.flash import flash_attention, flash_attention_seqlen

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--step", type=int, default=16)
ap.add_argument("--out", default="")
a = ap.parse_args()
DEV = "cuda:0"; dt = torch.bfloat16; Hq, Hkv, D = 8, 1, 128
flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)

def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay

def timed(fn, iters):
    run = graphed(fn)
    for _ in range(2): run()
    ts = []
    for _ in range(iters):
        flush.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = torch.tensor(ts)
    return t.mean().item(), t.std().item()

rows = ["| batch | prefix | suffix | hydragen us | no-sharing us | speed-up |", "|---|---|---|---|---|---|"]
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s: torch.randn(*s, device=DEV, dtype=dt, generator=g)
for B, P in ((512, 1024), (1024, 2048), (2048, 4096), (4096, 8192)):
    q, sk, sv = r(B, 1, Hq, D), r(1, P, Hkv, D), r(1, P, Hkv, D)
    Smax = 512
    k, v = r(B, Smax, Hkv, D), r(B, Smax, Hkv, D)
    kt = torch.cat([sk.expand(B, -1, -1, -1), k], 1).contiguous()
    vt = torch.cat([sv.expand(B, -1, -1, -1), v], 1).contiguous()
    for S in range(0, Smax + 1, a.step):
        lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
        tl = lens + P
        if S == 0:  # scripts/microbenchmark.py:76-83,110-120: no unique keys -> prefix-only / dense attention
            e = torch.empty(B, 0, Hkv, D, device=DEV, dtype=dt)
            hm, hs = timed(lambda: hydragen_attention_nopad(q, e, e, [sk], [sv]), a.iters)
        else:
            ks, vs = k[:, :S], v[:, :S]
            hm, hs = timed(lambda: hydragen_attention_nopad(q, ks, vs, [sk], [sv], seq_len=lens), a.iters)
        nm, ns = timed(lambda: flash_attention_seqlen(q, kt, vt, seq_len=tl), max(4, a.iters // 2))
        rows.append(f"| {B} | {P} | {S} | {hm:.1f} ± {hs:.1f} | {nm:.1f} ± {ns:.1f} | {nm / hm:.1f}x |")
        print(rows[-1], flush=True)
    del kt, vt, k, v
    torch.cuda.empty_cache()
if a.out:
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text("\n".join(rows) + "\n")
