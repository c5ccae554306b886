#!/usr/bin/env python3
"""Where the prefix pass's time goes, workgroup by workgroup (development tool; needs the ablation library):

    python tools/build_ablation.py
    HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so HYD_DBG=2048 python tools/prefix_timeline.py [--shape c2]

The decode operator of the bench (C2: B=1024, P=2048, 32/32 heads, bf16; a captured HIP graph of the one-call form, replayed
so that the prefix pass runs cold behind the suffix pass's 1 GiB stream, exactly as in bench.py's timed step) with the
stamped prefix kernel (prefix_unit_w64.h, ABL bit 11): every workgroup's wave 0 records s_memtime at six points and
s_memrealtime (100 MHz) at its first and last instruction, plus its XCC id.  Printed: the launch's extent in real time
(first start -> last end), per-XCD start skew, the per-part medians in real microseconds, the shader clock each workgroup
ran at (cycles / real time), and the kernel's duration by HIP events for the same replays."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from hydragen_amd import _lib
from hydragen_amd._lib import HYD_PHASE_ALL, HYD_PHASE_SHARED, HYD_PHASE_UNIQUE, DecodeParams
from hydragen_amd.attention import _fill_level
from hydragen_amd.flash import fill_suffix_params

DEV = "cuda:0"
SHAPES = {"c2": (1024, 2048, 64, 32, 32), "c5slice": (2048, 4096, 256, 8, 1), "c3": (64, 16384, 256, 32, 8)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="c2", choices=list(SHAPES))
    ap.add_argument("--replays", type=int, default=20)
    a = ap.parse_args()
    assert int(os.environ.get("HYD_DBG", "0")) & 2048, "run with the ablation library and HYD_DBG=2048"
    lib = _lib.load()
    B, P, S, Hq, Hkv = SHAPES[a.shape]
    D, dt = 128, torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(0)
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
    ws = torch.zeros(n + (1 << 20), dtype=torch.uint8, device=DEV)  # + room for the stamps behind the level's LSEs
    p.workspace, p.workspace_bytes = ws.data_ptr(), n
    stream_obj = torch.cuda.Stream()
    rows = B * Hq
    nsplit, grid = C.c_int32(), C.c_int32()
    from hydragen_amd._lib import PrefixParams
    pp = PrefixParams()
    pp.dtype, pp.B, pp.nq, pp.Hq, pp.Hkv, pp.D, pp.sb, pp.kv_len = p.suffix.dtype, B, 1, Hq, Hkv, D, 1, P
    pp.k_tok_stride = pp.v_tok_stride = sk.stride(1)
    lib.hyd_prefix_plan(C.byref(pp), C.byref(nsplit), C.byref(grid), None)
    al = lambda x: (x + 255) // 256 * 256
    if nsplit.value == 1:
        lse_off = al(rows * D * 2)
    else:
        lse_off = nsplit.value * al(rows * D * 4)
    # the kernel writes 16 words per workgroup behind the level's LSEs (one array, or nsplit padded ones)
    stamp_off = lse_off + (rows * 4 if nsplit.value == 1 else nsplit.value * al(rows * 4))

    with torch.cuda.stream(stream_obj):
        st = stream_obj.cuda_stream
        for ph in (HYD_PHASE_ALL,):
            p.phase = ph
            for _ in range(3):
                _lib.check(lib.hyd_decode_attn_fused(C.byref(p), st))
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream_obj):
            p.phase = HYD_PHASE_ALL
            _lib.check(lib.hyd_decode_attn_fused(C.byref(p), st))
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        # event-timed phases (eager) of the same shapes, for the kernel's own duration
        ev = []
        for _ in range(a.replays):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            p.phase = HYD_PHASE_SHARED
            _lib.check(lib.hyd_decode_attn_fused(C.byref(p), st))
            e1.record()
            p.phase = HYD_PHASE_UNIQUE
            _lib.check(lib.hyd_decode_attn_fused(C.byref(p), st))
            e2.record()
            torch.cuda.synchronize()
            ev.append((e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3))
        recs = []
        for _ in range(a.replays):
            gr.replay()
            torch.cuda.synchronize()
            w = ws[stamp_off:stamp_off + grid.value * 64].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
            recs.append(w.reshape(grid.value, 16).copy())
    ev = np.asarray(ev)
    print(f"# {a.shape}: B={B} P={P} S={S} {Hq}/{Hkv} heads; prefix plan {nsplit.value} splits x {grid.value // nsplit.value} units = {grid.value} workgroups")
    print(f"# eager phases by HIP events (median of {a.replays}): shared {np.median(ev[:, 0]):.1f} us, unique {np.median(ev[:, 1]):.1f} us")
    rows_out = []
    for w in recs:
        mt = w[:, 0:6]
        rt0, rt1 = w[:, 6], w[:, 7]
        d32 = lambda x, y: (x - y) & 0xffffffff
        t_first = rt0.min()
        start = d32(rt0, t_first) / 100.0          # us after the first workgroup's start
        end = d32(rt1, t_first) / 100.0
        life = d32(rt1, rt0) / 100.0
        cyc = d32(mt[:, 5], mt[:, 0]).astype(np.float64)
        parts = [d32(mt[:, i + 1], mt[:, i]).astype(np.float64) for i in range(5)]
        clock = cyc / np.maximum(life, 1e-3)        # MHz
        xcc = w[:, 8] & 0xf
        rows_out.append(dict(extent=end.max(), ramp=start.max(), start_med=np.median(start), life_med=np.median(life), life_max=life.max(),
                             life_min=life.min(), end_min=end.min(), clock_med=np.median(clock), clock_min=clock.min(), clock_max=clock.max(),
                             parts=[np.median(x) for x in parts], parts_max=[x.max() for x in parts],
                             xcd_start=[float(np.median(start[xcc == x])) if (xcc == x).any() else float("nan") for x in range(8)],
                             xcd_life=[float(np.median(life[xcc == x])) if (xcc == x).any() else float("nan") for x in range(8)],
                             xcd_end=[float(end[xcc == x].max()) if (xcc == x).any() else float("nan") for x in range(8)],
                             xcd_clock=[float(np.median(clock[xcc == x])) if (xcc == x).any() else float("nan") for x in range(8)],
                             xcd_loop=[float(np.median(parts[2][xcc == x])) if (xcc == x).any() else float("nan") for x in range(8)],
                             xcd_fixed=[float(np.median((parts[0] + parts[1] + parts[3] + parts[4])[xcc == x])) if (xcc == x).any() else float("nan") for x in range(8)]))
    med = lambda key: float(np.median([r_[key] for r_ in rows_out]))
    print(f"# stamps of {len(recs)} graph replays (medians over replays); real time = s_memrealtime at 100 MHz (10 ns steps)")
    print(f"launch extent, first workgroup start -> last workgroup end   {med('extent'):7.2f} us")
    print(f"  last workgroup START after the first (launch ramp)         {med('ramp'):7.2f} us   (median start {med('start_med'):.2f})")
    print(f"  workgroup lifetime  min / median / max                     {med('life_min'):7.2f} / {med('life_med'):.2f} / {med('life_max'):.2f} us")
    print(f"  first workgroup END                                        {med('end_min'):7.2f} us")
    print(f"  shader clock (cycles / real time) min / median / max       {med('clock_min'):7.0f} / {med('clock_med'):.0f} / {med('clock_max'):.0f} MHz")
    names = ["setup + Q / first K,V issue", "first wait + barrier (prologue burst)", "key loop", "key-half merge write + barrier", "normalise + stores (drained)"]
    pm = np.median([r_["parts"] for r_ in rows_out], axis=0)
    px = np.median([r_["parts_max"] for r_ in rows_out], axis=0)
    clk = med("clock_med")
    for nme, c_, x_ in zip(names, pm, px):
        print(f"  {nme:40s} median {c_:8.0f} cycles = {c_ / clk:6.2f} us   (slowest workgroup {x_:8.0f} = {x_ / clk:6.2f} us)")
    print("  per XCD: median start / median lifetime / last end (us): " +
          "  ".join(f"{x}: {s_:.2f}/{l_:.2f}/{e_:.2f}" for x, (s_, l_, e_) in enumerate(zip(
              np.median([r_["xcd_start"] for r_ in rows_out], axis=0), np.median([r_["xcd_life"] for r_ in rows_out], axis=0),
              np.median([r_["xcd_end"] for r_ in rows_out], axis=0)))))
    print("  per XCD: shader clock MHz / key-loop cycles / fixed-part cycles: " +
          "  ".join(f"{x}: {c_:.0f}/{l_:.0f}/{f_:.0f}" for x, (c_, l_, f_) in enumerate(zip(
              np.median([r_["xcd_clock"] for r_ in rows_out], axis=0), np.median([r_["xcd_loop"] for r_ in rows_out], axis=0),
              np.median([r_["xcd_fixed"] for r_ in rows_out], axis=0)))))
    del keep


if __name__ == "__main__":
    main()
