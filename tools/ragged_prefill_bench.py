#!/usr/bin/env python3
"""Timing of the paths that so far had parity only (development tool, GPU box; VERDICT r5 missing #4 / next #8):

  * a RAGGED shared level (packed K/V + cu_seqlens: `flash_attention_varlen`, /root/reference/hydragen/flash.py:309-351, reached
    through `hydragen_attention(..., use_varlens=[True])`, attention.py:282-338 -- which the reference itself documents as slow,
    attention.py:258-261) against its UNIFORM twin (same total keys, `[sb, P, Hkv, D]`), lengths +- 10 %;
  * the causal unique PREFILL pass (`flash_attention(q, k, v, causal=True)`, attention.py:343-345, llama.py:527-562) over 2048
    new tokens per sequence, with its fraction of the dense bf16 MFMA peak (causal flops = half of the square).

    python tools/ragged_prefill_bench.py [--iters 20]
HIP-graph replays, HIP events; back to back and behind a 512 MB flush (the reference's protocol)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hydragen_amd.attention import hydragen_attention
from hydragen_amd.flash import flash_attention

DEV = "cuda:0"
dt = torch.bfloat16


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def timed(g, iters, flush=None):
    for _ in range(3):
        g.replay()
    ts = []
    for _ in range(iters):
        if flush is not None:
            flush.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = torch.tensor(ts)
    return t.mean().item(), t.std().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, dtype=dt, generator=gen)  # noqa: E731
    print("## ragged shared level vs its uniform twin (decode, bf16, D = 128; us mean ± std: back to back | flushed)")
    print("| hierarchy | heads | uniform | ragged (+- 10 %, packed + cu_seqlens) | ragged / uniform (b2b) |")
    print("|---|---|---|---|---|")
    for name, B, S, Hq, Hkv, levels in (
        ("C4: 1 x 1024 + 32 x 64, 1024 sequences, suffix 32", 1024, 32, 32, 32, [(1, 1024), (32, 64)]),
        ("8 prompts x 2048, 1024 sequences, suffix 64", 1024, 64, 32, 32, [(8, 2048)]),
        ("32 prompts x 512, 2048 sequences, suffix 64 (8 q / 1 kv heads)", 2048, 64, 8, 1, [(32, 512)]),
    ):
        q, k, v = r(B, 1, Hq, 128), r(B, S, Hkv, 128), r(B, S, Hkv, 128)
        lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
        uk, uv, rk, rv, cus, maxs, usev = [], [], [], [], [], [], []
        for sb, P in levels:
            uk.append(r(sb, P, Hkv, 128))
            uv.append(r(sb, P, Hkv, 128))
            if sb == 1:  # one group cannot be ragged against itself: stays uniform in both twins
                rk.append(uk[-1]); rv.append(uv[-1]); cus.append(None); maxs.append(None); usev.append(False)
                continue
            # lengths P +- 10 %, alternating, same total
            d = max(1, P // 10)
            ls = [P + (d if i % 2 == 0 else -d) for i in range(sb)]
            cu = torch.tensor([0] + list(torch.tensor(ls).cumsum(0)), dtype=torch.int32, device=DEV)
            rk.append(r(sum(ls), Hkv, 128)); rv.append(r(sum(ls), Hkv, 128)); cus.append(cu); maxs.append(max(ls)); usev.append(True)
        n = len(levels)
        gu = capture(lambda: hydragen_attention(q, k, v, uk, uv, [None] * n, [None] * n, [False] * n, seq_lens=lens))
        gr = capture(lambda: hydragen_attention(q, k, v, rk, rv, cus, maxs, usev, seq_lens=lens))
        u0, u1, r0, r1 = timed(gu, a.iters), timed(gu, a.iters, flush), timed(gr, a.iters), timed(gr, a.iters, flush)
        print(f"| {name} | {Hq}/{Hkv} | {u0[0]:.1f} ± {u0[1]:.1f} \\| {u1[0]:.1f} ± {u1[1]:.1f} | {r0[0]:.1f} ± {r0[1]:.1f} \\| {r1[0]:.1f} ± {r1[1]:.1f} | {r0[0] / u0[0]:.2f} |",
              flush=True)
        del gu, gr
    print("\n## causal unique prefill: flash_attention(q, k, v, causal=True), bottom-right aligned, bf16, D = 128")
    print("| sequences x new tokens | heads | us (b2b) | us (flushed) | TFLOP/s (causal half) | of 2.5 PFLOP/s |")
    print("|---|---|---|---|---|---|")
    for b, nq, Hq, Hkv in ((8, 2048, 32, 32), (1, 2048, 32, 32), (8, 2048, 64, 8), (32, 512, 32, 32)):
        q, k, v = r(b, nq, Hq, 128), r(b, nq, Hkv, 128), r(b, nq, Hkv, 128)
        g = capture(lambda: flash_attention(q, k, v, causal=True))
        t0, t1 = timed(g, a.iters), timed(g, a.iters, flush)
        fl = 4.0 * b * Hq * nq * (nq + 1) / 2 * 128
        print(f"| {b} x {nq} | {Hq}/{Hkv} | {t0[0]:.1f} ± {t0[1]:.1f} | {t1[0]:.1f} ± {t1[1]:.1f} | {fl / t0[0] / 1e6:.0f} | {fl / t0[0] / 1e6 / 2500:.3f} |", flush=True)
        del g


if __name__ == "__main__":
    main()
