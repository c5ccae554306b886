#!/usr/bin/env python3
"""Config-level help for the whole-model ratio (no GEMM kernels are written here): PyTorch TunableOp
(hipBLASLt / rocBLAS solution selection) for the GEMM shapes of one Llama-2-7B decode step at M = batch 1024, tuned offline:

    python tools/tune_gemms.py --out gpurun_out/r06/tunableop_llama2_7b_m1024.csv

Prints per shape the default and the tuned time (HIP events, 50 launches) and what a decode step's GEMMs sum to (32 layers +
lm_head).  The results file is then used by
    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=<csv> python tools/bench_model.py --modes noattention
Shapes follow hydragen_amd/llama.py (q|k|v and gate|up run as ONE GEMM each over a fused weight; the reference runs the
same projections separately, /root/reference/hydragen/llama.py:440-470, scripts/synth.py:111-115 for the protocol)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/r06/tunableop_llama2_7b_m1024.csv")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--max-tuning-ms", type=int, default=50)
a = ap.parse_args()
DEV = "cuda:0"
M = a.batch
SHAPES = [  # name, K, N, launches per decode step
    ("q|k|v", 4096, 3 * 4096, 32),
    ("o_proj", 4096, 4096, 32),
    ("gate|up", 4096, 2 * 11008, 32),
    ("down", 11008, 4096, 32),
    ("lm_head", 4096, 32000, 1),
]
tun = torch.cuda.tunable


def time_linear(x, w, iters=50):
    for _ in range(5):
        F.linear(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        F.linear(x, w)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


Path(a.out).parent.mkdir(parents=True, exist_ok=True)
data = [(n, torch.randn(M, K, device=DEV, dtype=torch.bfloat16), torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.02, c) for n, K, N, c in SHAPES]
tun.enable(False)
base = [time_linear(x, w) for _, x, w, _ in data]
tun.enable(True)
tun.tuning_enable(True)
tun.set_filename(a.out)
try:
    tun.set_max_tuning_duration(a.max_tuning_ms)
    tun.set_max_tuning_iterations(100)
except Exception as ex:  # older API
    print("note:", ex)
for _, x, w, _ in data:
    F.linear(x, w)  # tunes this shape
torch.cuda.synchronize()
if hasattr(tun, "write_file"):
    tun.write_file(a.out)  # (else: written when the process exits, TunableOp's default)
tun.tuning_enable(False)
tuned = [time_linear(x, w) for _, x, w, _ in data]
print("| GEMM (M = %d) | K | N | default us | TunableOp us | TFLOP/s default -> tuned | launches / step |" % M)
print("|---|---|---|---|---|---|---|")
tb = tt = 0.0
for (n, K, N, c), b_, t_ in zip(SHAPES, base, tuned):
    fl = 2.0 * M * K * N
    print(f"| {n} | {K} | {N} | {b_:.1f} | {t_:.1f} | {fl / b_ / 1e6:.0f} -> {fl / t_ / 1e6:.0f} | {c} |")
    tb += b_ * c
    tt += t_ * c
print(f"\nGEMMs of one decode step (32 layers + lm_head): default {tb / 1e3:.2f} ms, tuned {tt / 1e3:.2f} ms")
print("results file:", a.out)
for row in tun.get_results():
    print("  ", row)
