#!/usr/bin/env python3
"""A/B of prefix-pass variants of the ablation library (development tool, GPU box):

    python tools/build_ablation.py
    HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so python tools/prefix_ab.py --dbg 0,8192,16384 [--shape c2]

For every HYD_DBG value (0 = the shipped instantiation): HYD_PHASE_SHARED timed back to back and INSIDE the step (behind
the suffix pass's HBM stream, as bench.py's timed step runs it), HIP events, min / median of `iters`; two passes over the
variants so that drift shows.  Also checks the variant's output against the shipped one's."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hydragen_amd import _lib
from hydragen_amd._lib import HYD_PHASE_SHARED, HYD_PHASE_UNIQUE, DecodeParams
from hydragen_amd.attention import _fill_level
from hydragen_amd.flash import fill_suffix_params

DEV = "cuda:0"
SHAPES = {"c2": (1024, 2048, 64, 32, 32), "c2tp8": (1024, 2048, 64, 4, 4), "c5slice": (2048, 4096, 256, 8, 1), "c3": (64, 16384, 256, 32, 8)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="c2", choices=list(SHAPES))
    ap.add_argument("--dbg", default="0,8192,16384")
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    lib = _lib.load()
    B, P, S, Hq, Hkv = SHAPES[a.shape]
    D, dt = 128, torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, dtype=dt, generator=g)  # noqa: E731
    q, kv, sk, sv = r(B, 1, Hq, D), r(2, B, S, Hkv, D), r(1, P, Hkv, D), r(1, P, Hkv, D)
    lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
    out = torch.empty_like(q)
    p = DecodeParams()
    keep = [fill_suffix_params(p.suffix, q, kv[0], kv[1], lens, out)]  # noqa: F841
    p.n_levels = 1
    _fill_level(p.levels[0], sk, sv, None, None, False, B)
    n = lib.hyd_decode_workspace_bytes(C.byref(p))
    ws = torch.zeros(n + (1 << 20), dtype=torch.uint8, device=DEV)
    p.workspace, p.workspace_bytes = ws.data_ptr(), n
    stream = torch.cuda.current_stream().cuda_stream

    def call(phase):
        p.phase = phase
        _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))

    def timed(in_step):
        ts = []
        for i in range(a.iters + 3):
            if in_step:
                call(HYD_PHASE_UNIQUE)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(HYD_PHASE_SHARED)
            e1.record()
            if in_step:
                call(HYD_PHASE_UNIQUE)
            torch.cuda.synchronize()
            if i >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        t = torch.tensor(ts)
        return t.min().item(), t.median().item()

    os.environ.pop("HYD_DBG", None)
    call(HYD_PHASE_SHARED)
    call(HYD_PHASE_UNIQUE)
    torch.cuda.synchronize()
    want = out.float().clone()
    vals = [int(x) for x in a.dbg.split(",")]
    print(f"# {a.shape}: B={B} P={P} S={S} {Hq}/{Hkv} heads; shared phase us min / median")
    print("| HYD_DBG | back to back | in step | max abs diff vs shipped |")
    print("|---|---|---|---|")
    for rep in range(2):
        for v in vals:
            os.environ["HYD_DBG"] = str(v)
            out.zero_()
            call(HYD_PHASE_SHARED)
            call(HYD_PHASE_UNIQUE)
            torch.cuda.synchronize()
            err = (out.float() - want).abs().max().item()
            b2b, ins = timed(False), timed(True)
            print(f"| {v} | {b2b[0]:6.2f} / {b2b[1]:6.2f} | {ins[0]:6.2f} / {ins[1]:6.2f} | {err:.1e} |", flush=True)
    os.environ.pop("HYD_DBG", None)


if __name__ == "__main__":
    main()
