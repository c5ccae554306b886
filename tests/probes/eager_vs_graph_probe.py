#!/usr/bin/env python3
"""Probe (GPU box): the one-call decode operator at C2, 200 launches back to back -- eager C calls against replays of its
captured HIP graph (what bench.py's timed steps are) -- us per step by one pair of events around the whole train."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from hydragen_amd import placement
from hydragen_amd.attention import hydragen_attention_nopad

DEV, dt = "cuda:0", torch.bfloat16
B, P, Hq, Hkv, D = 1024, 2048, 32, 32, 128
q, sk, sv = (torch.randn(*s, device=DEV, dtype=dt) for s in ((B, 1, Hq, D), (1, P, Hkv, D), (1, P, Hkv, D)))
(kv,), rep = placement.place_kv_arenas(1, (B, 128, Hkv, D), dt, DEV, Hq, zero=False)
kv.normal_()
for S in (16, 64, 128):
    lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
    fn = lambda: hydragen_attention_nopad(q, kv[0], kv[1], [sk], [sv], seq_len=lens)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    res = {}
    for rep_ in range(2):
        for name, run in (("eager", fn), ("graph", g.replay)):
            for _ in range(10): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / 200)
    print(f"S={S}: eager {res['eager']}  graph {res['graph']} us per step")
