"""One launch vs the prefix pass + suffix pass pair for small problems: us per call (graph replays), to place
kSingleLaunchMaxKeys (api.hip)."""
import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from hydragen_amd import attention as A, _lib

dev = "cuda:0"


def run(B, P, S, Hq, Hkv, D, single):
    q = torch.randn(B, 1, Hq, D, device=dev).half()
    k = torch.randn(B, S, Hkv, D, device=dev).half()
    v = torch.randn_like(k)
    sk = torch.randn(1, P, Hkv, D, device=dev).half()
    sv = torch.randn_like(sk)
    orig = A._launch_decode

    def launch(lib, p, ts, st):
        p.phase, p.shared_max_workgroups, p.single_launch_small = _lib.HYD_PHASE_ALL, 0, single
        _lib.check(lib.hyd_decode_attn_fused(C.byref(p), st))

    A._launch_decode = launch
    A._PARAM_CACHE.clear()
    try:
        f = lambda: A.hydragen_attention_nopad(q, k, v, [sk], [sv])
        for _ in range(3):
            f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 400 * 1e3
    finally:
        A._launch_decode = orig
        A._PARAM_CACHE.clear()


print("   B     P    S  Hq/Hkv   keys  one launch   two passes")
for B, P, S, Hq, Hkv in [(4, 64, 8, 4, 4), (8, 256, 16, 8, 1), (32, 256, 16, 8, 1), (32, 1024, 128, 8, 1), (2, 1000, 9, 16, 4),
                         (16, 512, 32, 8, 8), (64, 128, 16, 8, 8), (64, 512, 16, 8, 1), (128, 256, 16, 8, 1), (16, 1024, 64, 32, 8)]:
    D = 64 if Hq == 4 else 128
    a, b = run(B, P, S, Hq, Hkv, D, 1), run(B, P, S, Hq, Hkv, D, 0)
    print(f"{B:4d} {P:5d} {S:4d}  {Hq:2d}/{Hkv:<2d}  {B * Hkv * (P + S):6d}   {a:8.2f}    {b:8.2f}")
