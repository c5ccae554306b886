#!/usr/bin/env python3
"""Development tool: run one prefix-pass shape back to back for a few seconds and sample the GPU's shader clock and
socket power (rocm-smi) meanwhile -- the chip clocks to its power budget, so a kernel variant's wall time is
cycles / clock and the clock depends on what the variant does.
    python tools/clockwatch.py --P 8192 [--B 1024 --Hq 32 --Hkv 32] [--seconds 3]"""
import argparse, subprocess, sys, threading, time, re, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hydragen_amd import _lib
from hydragen_amd._lib import PrefixParams, HYD_LSE_BQH
from hydragen_amd.flash import _dtype_code

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=8192)
ap.add_argument("--B", type=int, default=1024)
ap.add_argument("--Hq", type=int, default=32)
ap.add_argument("--Hkv", type=int, default=32)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--zeros", action="store_true", help="all-zero q / k / v (the data-independent part of the power draw)")
a = ap.parse_args()
lib = _lib.load()
dev = "cuda:0"
dt = torch.bfloat16
mk = torch.zeros if a.zeros else torch.randn
q = mk(a.B, 1, a.Hq, a.D, device=dev, dtype=dt)
sk = mk(1, a.P, a.Hkv, a.D, device=dev, dtype=dt); sv = mk(1, a.P, a.Hkv, a.D, device=dev, dtype=dt)
out = torch.empty_like(q); lse = torch.zeros(a.B * a.Hq + 64, device=dev, dtype=torch.float32)
p = PrefixParams()
p.q, p.k, p.v, p.out, p.lse = q.data_ptr(), sk.data_ptr(), sv.data_ptr(), out.data_ptr(), lse.data_ptr()
p.k_group_stride, p.k_tok_stride, p.k_head_stride = sk.stride(0), sk.stride(1), sk.stride(2)
p.v_group_stride, p.v_tok_stride, p.v_head_stride = sv.stride(0), sv.stride(1), sv.stride(2)
p.dtype = _dtype_code(q); p.B, p.nq, p.Hq, p.Hkv, p.D = a.B, 1, a.Hq, a.Hkv, a.D
p.sb, p.kv_len, p.lse_layout, p.num_splits = 1, a.P, HYD_LSE_BQH, 0
n = lib.hyd_prefix_workspace_bytes(C.byref(p))
if n:
    ws = torch.empty(n, dtype=torch.uint8, device=dev); p.workspace, p.workspace_bytes = ws.data_ptr(), n
stream = torch.cuda.current_stream().cuda_stream
samples, stop = [], False
def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        m = re.search(r"sclk clock level: \w+: \((\d+)Mhz\)", r); w = re.search(r"Power \(W\): ([\d.]+)", r)
        samples.append((int(m.group(1)) if m else -1, float(w.group(1)) if w else -1.0))
th = threading.Thread(target=sampler); th.start()
for _ in range(20): _lib.check(lib.hyd_prefix_attn_fwd(C.byref(p), stream))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time(); iters = 0; e0.record()
while time.time() - t0 < a.seconds:
    for _ in range(200): _lib.check(lib.hyd_prefix_attn_fwd(C.byref(p), stream))
    iters += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
us = e0.elapsed_time(e1) * 1e3 / iters
fl = 4.0 * a.B * a.Hq * a.P * a.D
mid = samples[len(samples) // 3:] or samples
print(f"P={a.P} B={a.B}: {us:8.2f} us/launch  {fl/us/1e6:7.1f} TFLOP/s | sclk {[s[0] for s in mid][:12]} MHz  power {[s[1] for s in mid][:12]} W  ({len(samples)} samples)")
