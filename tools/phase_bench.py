#!/usr/bin/env python3
"""Per-phase timings of the decode operator on the split-KV shapes (development tool, GPU box):

    python tools/phase_bench.py [--iters 30] [--only C3]

For each shape: the one-call form (HYD_PHASE_ALL) and its two phases issued separately (HYD_PHASE_SHARED = the prefix
pass, HYD_PHASE_UNIQUE = the suffix pass + the merge with the prefix slices), each timed with HIP
events call by call, back to back and cold (512 MB written + 512 MB read before every call).  Also the plan of the
prefix pass (splits, grid)."""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hydragen_amd import _lib
from hydragen_amd._lib import HYD_PHASE_ALL, HYD_PHASE_SHARED, HYD_PHASE_UNIQUE, DecodeParams
from hydragen_amd.attention import _fill_level
from hydragen_amd.flash import fill_suffix_params

DEV = "cuda:0"
SHAPES = [
    # name, B, P, S, Hq, Hkv
    ("C3 (B=64,P=16384,S=256,32/8)", 64, 16384, 256, 32, 8),
    ("C5 slice (B=2048,P=4096,S=256,8/1)", 2048, 4096, 256, 8, 1),
    ("paper default (B=1024,P=2048,S=128,8/1)", 1024, 2048, 128, 8, 1),
    ("paper corner (B=32,P=1024,S=128,8/1)", 32, 1024, 128, 8, 1),
    ("C2 (B=1024,P=2048,S=128,32/32)", 1024, 2048, 128, 32, 32),
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
            clean.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = torch.tensor(ts)
    return t.median().item(), t.std().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--shape", action="append", default=[], help="extra shape B,P,S,Hq,Hkv (repeatable), e.g. a TP rank's shard of C2: 1024,2048,128,4,4")
    a = ap.parse_args()
    for sh in a.shape:
        B_, P_, S_, Hq_, Hkv_ = map(int, sh.split(","))
        SHAPES.append((f"custom (B={B_},P={P_},S={S_},{Hq_}/{Hkv_})", B_, P_, S_, Hq_, Hkv_))
    lib = _lib.load()
    flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)
    clean = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    stream = torch.cuda.current_stream().cuda_stream
    print("| shape | splits x units | whole b2b | whole cold | shared b2b | shared cold | unique b2b | unique cold |")
    print("|---|---|---|---|---|---|---|---|")
    for name, B, P, S, Hq, Hkv in SHAPES:
        if a.only and a.only not in name:
            continue
        D, dt = 128, torch.bfloat16
        r = lambda *s: torch.randn(*s, device=DEV, dtype=dt, generator=g)
        q, kv, sk, sv = r(B, 1, Hq, D), r(2, B, S, Hkv, D), r(1, P, Hkv, D), r(1, P, Hkv, D)
        k, v = kv[0], kv[1]  # one arena, K | V, as PerLayerKVCache allocates a layer's unique caches
        lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
        out = torch.empty_like(q)
        p = DecodeParams()
        keep = [fill_suffix_params(p.suffix, q, k, v, lens, out)]
        p.n_levels = 1
        _fill_level(p.levels[0], sk, sv, None, None, False, B)
        n = lib.hyd_decode_workspace_bytes(C.byref(p))
        ws = torch.empty(max(n, 1), dtype=torch.uint8, device=DEV)
        p.workspace, p.workspace_bytes = ws.data_ptr(), n

        from hydragen_amd._lib import PrefixParams
        pp = PrefixParams()
        pp.dtype, pp.B, pp.nq, pp.Hq, pp.Hkv, pp.D, pp.sb, pp.kv_len = p.suffix.dtype, B, 1, Hq, Hkv, D, 1, P
        pp.k_tok_stride = pp.v_tok_stride = sk.stride(1)
        ns, grid = C.c_int32(), C.c_int32()
        lib.hyd_prefix_plan(C.byref(pp), C.byref(ns), C.byref(grid), None)

        def call(phase):
            def f():
                p.phase = phase
                _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))
            return f

        cells = []
        for ph in (HYD_PHASE_ALL, HYD_PHASE_SHARED, HYD_PHASE_UNIQUE):
            if ph == HYD_PHASE_UNIQUE:
                call(HYD_PHASE_SHARED)()  # the unique phase expects the workspace as the shared phase left it
            m, s_ = timed(call(ph), a.iters)
            cm, cs = timed(call(ph), a.iters, flush, clean)
            cells += [f"{m:7.1f} ± {s_:4.1f}", f"{cm:7.1f} ± {cs:4.1f}"]
        print(f"| {name} | {ns.value} x {grid.value // max(ns.value, 1)} | " + " | ".join(cells) + " |", flush=True)
        del keep


if __name__ == "__main__":
    main()
