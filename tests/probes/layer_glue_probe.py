"""Timing of the decoder-layer glue at batch 1024 (Llama-2-7B widths): torch add + rms_norm vs hyd_add_rmsnorm, torch
silu * mul on split views vs hyd_swiglu, and the residual folded into the GEMM (addmm) vs a separate add."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from hydragen_amd import layer_ops

dev = "cuda:0"
B, H, I = 1024, 4096, 11008
x = torch.randn(B, 1, H, device=dev).bfloat16()
r = torch.randn(B, 1, H, device=dev).bfloat16()
w = torch.ones(H, device=dev).bfloat16()
gu = torch.randn(B, 1, 2 * I, device=dev).bfloat16()
g, u = gu.split(I, dim=-1)
Wo = (0.02 * torch.randn(H, H, device=dev)).bfloat16()
Wd = (0.02 * torch.randn(H, I, device=dev)).bfloat16()
act = torch.randn(B, I, device=dev).bfloat16()
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timeit(f, n=50):
    for _ in range(5):
        f()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            f()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / 10 * 1e3


print("torch add + rms_norm      %.2f us" % timeit(lambda: torch.nn.functional.rms_norm(x + r, (H,), w, 1e-5)))
print("hyd_add_rmsnorm           %.2f us" % timeit(lambda: layer_ops.add_rms_norm(x, r, w, 1e-5)))
print("torch rms_norm alone      %.2f us" % timeit(lambda: torch.nn.functional.rms_norm(x, (H,), w, 1e-5)))
print("hyd rmsnorm alone         %.2f us" % timeit(lambda: layer_ops.add_rms_norm(x, None, w, 1e-5)))
print("torch silu * up (views)   %.2f us" % timeit(lambda: torch.nn.functional.silu(g) * u))
print("hyd_swiglu                %.2f us" % timeit(lambda: layer_ops.swiglu(g, u)))
x2, r2 = x.view(B, H), r.view(B, H)
print("o_proj linear             %.2f us" % timeit(lambda: torch.nn.functional.linear(x2, Wo)))
print("o_proj addmm(residual)    %.2f us" % timeit(lambda: torch.addmm(r2, x2, Wo.t())))
print("down linear               %.2f us" % timeit(lambda: torch.nn.functional.linear(act, Wd)))
print("down addmm(residual)      %.2f us" % timeit(lambda: torch.addmm(r2, act, Wd.t())))
