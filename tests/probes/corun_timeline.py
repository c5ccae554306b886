"""Timeline of the co-run probe: when do X (MFMA + VALU probe kernel on a side stream) and Y (the unique phase on the
main stream) start and end, relative to a common event?   python tests/probes/corun_timeline.py <big 0..4> [lds]"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import bench
dev = "cuda:0"; dt = torch.bfloat16
lib = C.CDLL("build_probe/libcorun.so")
lib.corun_launch_cfg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
BIG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
LDS = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
B, P, S, H, D = 1024, 2048, 128, 32, 128
q = torch.randn(B, 1, H, D, device=dev, dtype=dt); sk = torch.randn(1, P, H, D, device=dev, dtype=dt); sv = torch.randn_like(sk)
k = torch.randn(B, S, H, D, device=dev, dtype=dt); v = torch.randn_like(k)
ops = bench.Ops(q, k, v, sk, sv, [128])
out = torch.zeros(16, device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
E = lambda: torch.cuda.Event(enable_timing=True)
rows = []
for rep in range(8):
    e0, xs, xe, ys, ye = E(), E(), E(), E(), E()
    e0.record(main); side.wait_event(e0)
    xs.record(side); lib.corun_launch_cfg(out.data_ptr(), 400, 1, side.cuda_stream, BIG, LDS); xe.record(side)
    ys.record(main); ops.unique_phase(128, main.cuda_stream); ye.record(main)
    main.wait_event(xe); torch.cuda.synchronize()
    rows.append([e0.elapsed_time(t) * 1e3 for t in (xs, xe, ys, ye)])
r = torch.tensor(rows[3:]).mean(0)
print(f"big={BIG}: X start {r[0]:6.1f} end {r[1]:6.1f} | Y start {r[2]:6.1f} end {r[3]:6.1f}  (us after the common event)")
