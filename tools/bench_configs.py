#!/usr/bin/env python3
"""Operator-level timings of every BASELINE config, on one MI355X:

    python tools/bench_configs.py [--iters 30] [--out gpurun_out/configs.md]

For each config: the Hydragen operator (`hydragen_attention_nopad`, fused decode entry point) and the
no-sharing baseline (`flash_attention_seqlen` over a private [P+S] KV per sequence -- the reference's
`go_baseline` with --unique-seq-len, scripts/microbenchmark.py:91-127), timed per call with HIP events
after a separate warm-up (hydragen/benchmark_utils.py:82-137), (a) back to back and (b) with a 512 MB
buffer written between calls so that neither L2 nor the 256 MB Infinity Cache holds the shared KV
(scripts/microbenchmark.py:28-47).  Reports mean / std / rstd and flags rstd > 10 % (scripts/synth.py:240-245).
Also the reference microbenchmark's own default head configuration (Hq=8, Hkv=1, D=128,
scripts/microbenchmark.py:136-138) at the (B, P) corners of the paper sweep (docs/sweeps_from_paper.md:159-161).
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hydragen_amd.attention import hydragen_attention_nopad
from hydragen_amd.flash import flash_attention_seqlen

DEV = "cuda:0"
CONFIGS = [
    # name, B, shared levels [(sb, P)], S, Hq, Hkv, D, dtype
    ("C1 literal (B=4,P=64,S=8,4h,D=64) fp16", 4, [(1, 64)], 8, 4, 4, 64, torch.float16),
    ("C2 (B=1024,P=2048,S=128,32/32)", 1024, [(1, 2048)], 128, 32, 32, 128, torch.bfloat16),
    ("C2 at S=16", 1024, [(1, 2048)], 16, 32, 32, 128, torch.bfloat16),
    ("C3 (B=64,P=16384,S=256,32/8)", 64, [(1, 16384)], 256, 32, 8, 128, torch.bfloat16),
    ("C4 two-level (1x1024 + 32x64, B=1024,S=32)", 1024, [(1, 1024), (32, 64)], 32, 32, 32, 128, torch.bfloat16),
    ("C5 TP=8 slice (B=2048,P=4096,S=256,8/1)", 2048, [(1, 4096)], 256, 8, 1, 128, torch.bfloat16),
    ("C5 whole job (B=2048,P=4096,S=256,64/8)", 2048, [(1, 4096)], 256, 64, 8, 128, torch.bfloat16),
    ("paper sweep corner (B=32,P=1024,S=128,8/1)", 32, [(1, 1024)], 128, 8, 1, 128, torch.bfloat16),
    ("paper sweep corner (B=2048,P=16256,S=128,8/1)", 2048, [(1, 16256)], 128, 8, 1, 128, torch.bfloat16),
]


def timed(fn, iters, flush=None, clean=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        if flush is not None:
            flush.add_(1)
        if clean is not None:
            clean.sum()  # read-only pass: evicts the flush's dirty lines, whose write-back the timed call would pay
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = torch.tensor(ts)
    return t.mean().item(), t.std().item()


def rel_l2(out, q, k, v, sks, svs, S, nseq=8):
    """Relative L2 error of `out` on `nseq` sequences (all heads) against fp64 softmax attention over the concatenated
    [level 0; level 1; ...; unique] keys of the same rounded inputs (sequence b uses group b // (B / sb) of every level,
    /root/reference/hydragen/attention.py:264-268)."""
    B, _, Hq, D = q.shape
    g = Hq // k.shape[2]
    idx = torch.linspace(0, B - 1, min(nseq, B), device=q.device).long()
    ks = [sk[idx // (B // sk.shape[0])] for sk in sks] + [k[idx, :S]]
    vs = [sv[idx // (B // sv.shape[0])] for sv in svs] + [v[idx, :S]]
    kk = torch.cat(ks, 1).double().repeat_interleave(g, 2)
    vv = torch.cat(vs, 1).double().repeat_interleave(g, 2)
    sc = torch.einsum("bqhd,bkhd->bhqk", q[idx].double(), kk) / D ** 0.5
    want = torch.einsum("bhqk,bkhd->bqhd", sc.softmax(-1), vv)
    got = out[idx].double()
    return float((got - want).norm() / want.norm()), float((got - want).abs().max() / want.abs().max())


def fmt(m, s):
    r = s / m if m > 0 else 0.0
    return f"{m:9.1f} ± {s:5.1f}{' (rstd>10%!)' if r > 0.10 else ''}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="", help="run only the configs whose name contains this (e.g. 'C5')")
    ap.add_argument("--no-baseline", action="store_true", help="skip the no-sharing leg (for kernel profiles)")
    a = ap.parse_args()
    flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)
    clean = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)
    lines = [
        "Flushed = 512 MB written before every call (the reference's protocol); the write leaves up to 256 MB of dirty lines in",
        "the Infinity Cache whose write-back lands in the timed call.  Cold = the same flush followed by a 512 MB read-only pass:",
        "nothing of the call's data is cached and nothing is waiting to be written back.",
        "",
        "rel L2 / max = relative L2 error and max |error| / max |exact| of the operator's output on 8 sequences x all heads against fp64 attention",
        "over the concatenated keys of the same rounded inputs (one bf16 output rounding alone: 1.1e-3 / <= 2^-9 = 2.0e-3).",
        "",
        "| config | hydragen us (back to back) | hydragen us (flushed) | hydragen us (cold) | no-sharing us (flushed) | speed-up (flushed) | hydragen KV bytes | effective GB/s (cold) | rel L2 / max |",
        "|---|---|---|---|---|---|---|---|---|",
    ]
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, B, levels, S, Hq, Hkv, D, dt in CONFIGS:
        if a.only and a.only not in name:
            continue
        r = lambda *s: torch.randn(*s, device=DEV, dtype=dt, generator=g)
        q, kv = r(B, 1, Hq, D), r(2, B, S, Hkv, D)
        k, v = kv[0], kv[1]  # one arena, K | V, as PerLayerKVCache allocates a layer's unique caches
        sks, svs = [r(sb, P, Hkv, D) for sb, P in levels], [r(sb, P, Hkv, D) for sb, P in levels]
        lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
        hyd = lambda: hydragen_attention_nopad(q, k, v, sks, svs, seq_len=lens)
        l2, mx = rel_l2(hyd(), q, k, v, sks, svs, S)
        hm, hs = timed(hyd, a.iters)
        fm, fs = timed(hyd, a.iters, flush)
        cm, cs = timed(hyd, a.iters, flush, clean)
        e = q.element_size()
        kv_bytes = 2 * e * Hkv * D * (B * S + sum(sb * P for sb, P in levels)) + 2 * B * Hq * D * e
        ptot = sum(P for _, P in levels)
        ns_bytes = 2 * e * B * (ptot + S) * Hkv * D
        nm = ns = None
        if ns_bytes < 150e9 and not a.no_baseline:
            try:
                per = [B // sb for sb, _ in levels]
                kt = torch.cat([sk.repeat_interleave(p, 0) for sk, p in zip(sks, per)] + [k], 1).contiguous()
                vt = torch.cat([sv.repeat_interleave(p, 0) for sv, p in zip(svs, per)] + [v], 1).contiguous()
                tl = lens + ptot
                nm, ns = timed(lambda: flash_attention_seqlen(q, kt, vt, seq_len=tl), max(5, a.iters // 3), flush)
                del kt, vt
            except torch.OutOfMemoryError:
                nm = None
        torch.cuda.empty_cache()
        sp = f"{nm / fm:5.1f}x" if nm else "n/a (KV > HBM budget)"
        nstr = fmt(nm, ns) if nm else f"({ns_bytes / 1e9:.0f} GB of KV)"
        lines.append(f"| {name} | {fmt(hm, hs)} | {fmt(fm, fs)} | {fmt(cm, cs)} | {nstr} | {sp} | {kv_bytes / 2**20:.0f} MiB | {kv_bytes / cm / 1e3:.0f} | {l2:.2e} / {mx:.2e} |")
        print(lines[-1], flush=True)
    txt = "\n".join(lines) + "\n"
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(txt)


if __name__ == "__main__":
    main()
