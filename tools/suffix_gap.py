#!/usr/bin/env python3
"""Where the dominant kernel's time goes at C2 (development tool, GPU box):

    python tools/suffix_gap.py [--iters 30] [--legs sweep,instep,occ]

`sweep`   suffix pass alone over S, with and without the fused merge epilogue (n_partials 1 / 0), on (i) a placed
          128-row K|V arena (the bench's: 1 MiB between sequences), (ii) the SAME sequences in a 2176-row arena
          (17 MiB between sequences: the no-sharing leg's), rows up to 2176: splits "row length" from "cache stride"
          from "merge epilogue".
`instep`  HYD_PHASE_UNIQUE timed alone (back to back; one by one behind an idle gap) against the same launch behind
          HYD_PHASE_SHARED (events as bench.py places them) and behind an 8 MiB write burst alone.
`occ`     occupancy A/B of the ablation library (HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so): HYD_SUFFIX_OCC =
          <key iterations in flight><waves per SIMD>.
All times are HIP events on the launch stream; every figure is min / median of `iters` launches."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hydragen_amd import _lib, placement
from hydragen_amd._lib import HYD_PHASE_SHARED, HYD_PHASE_UNIQUE, DecodeParams, SuffixParams
from hydragen_amd.attention import _fill_level
from hydragen_amd.flash import fill_suffix_params

DEV = "cuda:0"
B, Hq, Hkv, D, P = 1024, 32, 32, 128, 2048
dt = torch.bfloat16


def alg_bytes(S, np_):
    return 2 * 2 * Hkv * D * B * S + (2 + np_) * B * Hq * D * 2 + 4 * B * Hq * (np_ + 0)


def times(fn, iters, pre=None, sync_each=False):
    for _ in range(3):
        if pre:
            pre()
        fn()
    torch.cuda.synchronize()
    ts = []
    if sync_each or pre:
        for _ in range(iters):
            if pre:
                pre()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
    else:  # back to back: events between consecutive launches
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        evs[0].record()
        for i in range(iters):
            fn()
            evs[i + 1].record()
        torch.cuda.synchronize()
        ts = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(iters)]
    t = torch.tensor(ts)
    return t.min().item(), t.median().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--shape", default="", help="B,Hq,Hkv instead of C2's 1024,32,32 (legs that take the suffix pass alone: interleave, capacity, stride)")
    ap.add_argument("--caps", default="128,144", help="interleave leg: cache rows")
    ap.add_argument("--layouts", default="2,B;B,2", help="interleave leg: K|V arena layouts, of 2,B  B,2  B,r,2  B,r,H,2 (semicolon-separated)")
    ap.add_argument("--legs", default="sweep,instep,occ")
    ap.add_argument("--occ", default="0,48,38,28,65,84")
    ap.add_argument("--rows", default="0:8:1,1:8:1,1:8:0,1:4:1,1:4:0")
    ap.add_argument("--S", default="16,32,64,128")
    a = ap.parse_args()
    global B, Hq, Hkv
    if a.shape:
        B, Hq, Hkv = (int(x) for x in a.shape.split(","))
    legs = a.legs.split(",")
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=DEV).manual_seed(0)
    q = torch.randn(B, 1, Hq, D, device=DEV, dtype=dt, generator=g)
    out = torch.empty_like(q)
    pout = torch.randn(B, 1, Hq, D, device=DEV, dtype=dt, generator=g)
    plse = torch.randn(B, 1, Hq, device=DEV, dtype=torch.float32, generator=g)
    if set(legs) & {"sweep", "instep", "occ", "rows"}:
        arenas, rep = placement.place_kv_arenas(1, [B, 128, Hkv, D], dt, DEV, Hq, zero=False)
        a128 = arenas[0]
        a128.normal_()
        print("placement:", rep, flush=True)

    def suffix_call(arena, S, np_):
        lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
        sp = SuffixParams()
        keep = fill_suffix_params(sp, q, arena[0], arena[1], lens, out)
        sp.n_partials = np_
        if np_:
            sp.partials[0].out, sp.partials[0].lse, sp.partials[0].count, sp.partials[0].is_f32 = pout.data_ptr(), plse.data_ptr(), 1, 0
        return (lambda: _lib.check(lib.hyd_suffix_attn_fwd(C.byref(sp), stream))), (keep, lens, sp)

    if "sweep" in legs:
        print("\n## length sweep (us min / median; TB/s of the median; algorithmic bytes)")
        print("| arena rows | S | n_partials=0 | TB/s | n_partials=1 | TB/s |")
        print("|---|---|---|---|---|---|")
        for S in (16, 32, 48, 64, 96, 128):
            cells = []
            for np_ in (0, 1):
                fn, keep = suffix_call(a128, S, np_)
                mn, md = times(fn, a.iters)
                cells += [f"{mn:7.1f} / {md:7.1f}", f"{alg_bytes(S, np_) / md / 1e6:5.2f}"]
            print(f"| 128 | {S} | " + " | ".join(cells) + " |", flush=True)
        big = torch.empty((2, B, 2176, Hkv, D), dtype=dt, device=DEV)
        big.normal_()
        for S in (16, 64, 128, 256, 512, 1024, 2176):
            cells = []
            for np_ in (0, 1):
                fn, keep = suffix_call(big, S, np_)
                mn, md = times(fn, max(5, a.iters // (1 + S // 256)))
                cells += [f"{mn:7.1f} / {md:7.1f}", f"{alg_bytes(S, np_) / md / 1e6:5.2f}"]
            print(f"| 2176 | {S} | " + " | ".join(cells) + " |", flush=True)
        del big
        torch.cuda.empty_cache()

    def decode_params(S):
        lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
        sk = torch.randn(1, P, Hkv, D, device=DEV, dtype=dt, generator=g)
        sv = torch.randn(1, P, Hkv, D, device=DEV, dtype=dt, generator=g)
        p = DecodeParams()
        keep = [fill_suffix_params(p.suffix, q, a128[0], a128[1], lens, out), lens, sk, sv]
        p.n_levels = 1
        _fill_level(p.levels[0], sk, sv, None, None, False, B)
        n = lib.hyd_decode_workspace_bytes(C.byref(p))
        ws = torch.empty(max(n, 1), dtype=torch.uint8, device=DEV)
        p.workspace, p.workspace_bytes = ws.data_ptr(), n
        keep.append(ws)

        def call(phase):
            def f():
                p.phase = phase
                _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))
            return f
        return call, keep

    if "instep" in legs:
        print("\n## HYD_PHASE_UNIQUE alone vs inside the step (us min / median)")
        print("| S | unique b2b | unique one by one (sync before) | unique behind shared (event between) | unique behind an 8 MiB write | unique behind 512 MB flush | shared b2b |")
        print("|---|---|---|---|---|---|---|")
        burst_src = torch.randn(B, 1, Hq, D, device=DEV, dtype=dt)
        burst_dst = torch.empty_like(burst_src)
        flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)
        for S in (32, 64, 128):
            call, keep = decode_params(S)
            call(HYD_PHASE_SHARED)()
            u, s = call(HYD_PHASE_UNIQUE), call(HYD_PHASE_SHARED)
            c0 = times(u, a.iters)
            c1 = times(u, a.iters, sync_each=True)
            c2 = times(u, a.iters, pre=s)
            c3 = times(u, a.iters, pre=lambda: burst_dst.copy_(burst_src))
            c4 = times(u, a.iters, pre=lambda: flush.add_(1))
            c5 = times(s, a.iters)
            print(f"| {S} | " + " | ".join(f"{x[0]:6.1f} / {x[1]:6.1f}" for x in (c0, c1, c2, c3, c4, c5)) + " |", flush=True)

    _extra_keys = set()

    def set_variant(v):
        """occ leg: HYD_SUFFIX_OCC value; rows legs: 'ROWS:UT:ROT[:PIPE]' (token-row kernel on/off, tokens in flight, rotated start, pipelined form)."""
        for k in ("HYD_SUFFIX_OCC", "HYD_SUFFIX_ROWS", "HYD_ROWS_UT", "HYD_ROWS_ROT", "HYD_ROWS_PIPE"):
            os.environ.pop(k, None)
        for k in [k for k in os.environ if k in _extra_keys]:
            os.environ.pop(k)
        if v is None:
            return
        if ";" in v:  # 'ROWS:UT:ROT:PIPE;NAME=value;...': further development switches of the ablation library (e.g. HYD_ROWS_TS=1)
            v, *extra = v.split(";")
            for kv in extra:
                k_, val = kv.split("=")
                os.environ[k_] = val
                _extra_keys.add(k_)
        if ":" in v:
            r_, u_, o_, *p_ = v.split(":")  # optional 4th field (HYD_ROWS_PIPE): 0 = the first form (loads, then arithmetic), 1 / 2 = blind K requests, 3 = the product form
            os.environ.update(HYD_SUFFIX_ROWS=r_, HYD_ROWS_UT=u_, HYD_ROWS_ROT=o_, HYD_ROWS_PIPE=p_[0] if p_ else "0")
        else:
            os.environ.update(HYD_SUFFIX_ROWS="0", HYD_SUFFIX_OCC=v)

    for leg, default in (("occ", a.occ), ("rows", a.rows)):
        if leg not in legs:
            continue
        print(f"\n## {leg} A/B (ablation library only; first column = reference) -- unique phase b2b, us min / median; [max |diff| against the first column's output]")
        vals = default.split(",")
        print("| S | " + " | ".join(vals) + " |")
        print("|---|" + "---|" * len(vals))
        for S in [int(x) for x in a.S.split(",")]:
            call, keep = decode_params(S)
            call(HYD_PHASE_SHARED)()
            u = call(HYD_PHASE_UNIQUE)
            set_variant(vals[0])
            u()
            torch.cuda.synchronize()
            want = out.float().clone()
            cells = []
            for rep_ in range(2):  # two passes over the variants: drift shows
                row = []
                for o in vals:
                    set_variant(o)
                    out.zero_()
                    u()
                    torch.cuda.synchronize()
                    err = (out.float() - want).abs().max().item()
                    mn, md = times(u, a.iters)
                    row.append(f"{mn:6.1f} / {md:6.1f} [{err:.1e}]")
                cells.append(row)
            set_variant(None)
            for row in cells:
                print(f"| {S} | " + " | ".join(row) + " |", flush=True)

    if "capacity" in legs:
        print("\n## cache capacity (= rows between sequences) x kernel: suffix pass alone, n_partials 1, plain allocations; us median (TB/s)")
        vals = a.rows.split(",")
        print("| cache rows | S | " + " | ".join(vals) + " |")
        print("|---|---|" + "---|" * len(vals))
        for cap in (128, 144, 256, 512, 1024, 2176):
            arena = torch.empty((2, B, cap, Hkv, D), dtype=dt, device=DEV)
            arena.normal_()
            for S in (64, 128, 512, 2176):
                if S > cap:
                    continue
                fn, keep = suffix_call(arena, S, 1)
                row = []
                for o in vals:
                    set_variant(o)
                    mn, md = times(fn, max(5, a.iters // (1 + S // 256)))
                    row.append(f"{md:7.1f} ({alg_bytes(S, 1) / md / 1e6:4.2f})")
                set_variant(None)
                print(f"| {cap} | {S} | " + " | ".join(row) + " |", flush=True)
            del arena
            torch.cuda.empty_cache()

    if "interleave" in legs:
        print("\n## K|V arena layout: [2, B, rows, Hkv, D] (K of all sequences, then V: sequences `rows` token rows apart) against [B, 2, rows, Hkv, D] "
              "(a sequence's K then its V: sequences 2 x rows apart at no extra memory); suffix pass alone, n_partials 1, plain allocations, "
              "three fresh allocations of each, alternating; us median")
        vals = a.rows.split(",")
        print("| cache rows | layout | alloc | " + " | ".join(f"S=rows/{d_} {v}" for d_ in (4, 2, 1) for v in vals) + " |")
        print("|---|---|---|" + "---|" * (3 * len(vals)))
        for cap in [int(x) for x in a.caps.split(",")]:
            held = []
            for rep_ in range(3):
                for layout in a.layouts.split(";"):
                    if layout == "2,B":
                        arena = torch.empty((2, B, cap, Hkv, D), dtype=dt, device=DEV)
                        kk, vv = arena[0], arena[1]
                    elif layout == "B,2":
                        arena = torch.empty((B, 2, cap, Hkv, D), dtype=dt, device=DEV)
                        kk, vv = arena[:, 0], arena[:, 1]
                    elif layout == "B,r,2":  # a token's K row, then its V row (16 KB contiguous per token at C2)
                        arena = torch.empty((B, cap, 2, Hkv, D), dtype=dt, device=DEV)
                        kk, vv = arena[:, :, 0], arena[:, :, 1]
                    else:  # "B,r,H,2": a head's K row, then its V row (512 B contiguous per token and head)
                        arena = torch.empty((B, cap, Hkv, 2, D), dtype=dt, device=DEV)
                        kk, vv = arena[:, :, :, 0], arena[:, :, :, 1]
                    arena.normal_()
                    held.append(arena)  # keep it: the next allocation lands somewhere else
                    cells = []
                    for S in (cap // 4, cap // 2, cap):
                        fn, keep = suffix_call((kk, vv), S, 1)
                        for o in vals:
                            set_variant(o)
                            cells.append("%6.1f" % times(fn, a.iters)[1])
                        set_variant(None)
                    print(f"| {cap} | [{layout}] | {rep_} | " + " | ".join(cells) + " |", flush=True)
            del held
            torch.cuda.empty_cache()

    if "stride" in legs:
        print("\n## batch stride of the K|V caches (bytes between sequences; 128 token rows = 1 MiB used per sequence and tensor), suffix pass alone, "
              "n_partials 1, plain allocation; us median per variant " + a.rows)
        vals = a.rows.split(",")
        row_el = Hkv * D
        pads = [k * 512 for k in range(0, 33)] + [24 << 10, 32 << 10, 48 << 10, 64 << 10, 96 << 10, 128 << 10, 192 << 10, 256 << 10, 384 << 10, 512 << 10]
        strides = [128 * row_el * 2 + p_ for p_ in pads] + [r_ * row_el * 2 for r_ in (160, 192, 224, 256, 320, 384, 448, 512, 640, 768, 1024, 1536, 2048, 2176)]
        print("| stride bytes | = rows + pad | " + " | ".join(f"S=64 {v}" for v in vals) + " | " + " | ".join(f"S=128 {v}" for v in vals) + " |")
        print("|---|---|" + "---|" * (2 * len(vals)))
        for sb in strides:
            bs = sb // 2  # elements
            flat = torch.empty(2 * B * bs, device=DEV, dtype=dt)
            flat.normal_()
            kk = flat[: B * bs].as_strided((B, 128, Hkv, D), (bs, row_el, D, 1))
            vv = flat[B * bs:].as_strided((B, 128, Hkv, D), (bs, row_el, D, 1))
            cells = []
            for S in (64, 128):
                fn, keep = suffix_call((kk, vv), S, 1)
                for o in vals:
                    set_variant(o)
                    cells.append("%6.1f" % times(fn, a.iters)[1])
                set_variant(None)
            print(f"| {sb} | {sb // (row_el * 2)} rows + {sb % (row_el * 2)} | " + " | ".join(cells) + " |", flush=True)
            del flat, kk, vv


if __name__ == "__main__":
    main()
