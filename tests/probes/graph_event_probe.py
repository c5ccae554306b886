#!/usr/bin/env python3
"""Probe (GPU box): can HIP events recorded INSIDE a captured graph time the kernels of a replay on this stack?
torch.cuda.Event(enable_timing=True, external=True) maps to hipEventRecordWithFlags(hipEventRecordExternal)."""
import torch

x = torch.randn(64 << 20, device="cuda:0")
y = torch.empty_like(x)
try:
    ev = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(3)]
except TypeError as ex:
    print("no external events in this torch:", ex)
    raise SystemExit(0)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y.copy_(x); y.mul_(2.0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        ev[0].record()
        y.copy_(x)
        ev[1].record()
        y.mul_(2.0)
        ev[2].record()
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        print("replay", i, "copy us", ev[0].elapsed_time(ev[1]) * 1e3, "mul us", ev[1].elapsed_time(ev[2]) * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("whole replay by outside events us", e0.elapsed_time(e1) * 1e3)
except Exception as ex:  # noqa: BLE001
    print("in-graph events do not work here:", type(ex).__name__, ex)
