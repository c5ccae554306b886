import ctypes as C, sys, torch
sys.path.insert(0, ".")
import bench
dev = "cuda:0"; dt = torch.bfloat16
# build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tests/probes/corun_probe.hip -o build_probe/libcorun.so
# run from the repo root on the GPU box:  python tests/probes/corun_exp.py [big 0|1] [lds bytes]
lib = C.CDLL("build_probe/libcorun.so")
lib.corun_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.corun_launch_small.argtypes = lib.corun_launch.argtypes
lib.corun_launch_cfg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
BIG = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # 1: 251 registers per wave, 0: ~50
LDS = int(sys.argv[2]) if len(sys.argv) > 2 else 131072  # bytes of LDS per workgroup
launch = lambda o, it, nv, st: lib.corun_launch_cfg(o, it, nv, st, BIG, LDS)
B, P, S, H, D = 1024, 2048, 128, 32, 128
q = torch.randn(B, 1, H, D, device=dev, dtype=dt); sk = torch.randn(1, P, H, D, device=dev, dtype=dt); sv = torch.randn_like(sk)
k = torch.randn(B, S, H, D, device=dev, dtype=dt); v = torch.randn_like(k)
lens = [64, 128]
ops = bench.Ops(q, k, v, sk, sv, lens)
out = torch.zeros(16, device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main); fn(); e1.record(main); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    x = torch.tensor(ts); return f"{x.mean():7.1f}"
for iters in (400,):
    X = lambda st: launch(out.data_ptr(), iters, 1, st)
    line = f"iters={iters}: X alone {t(lambda: X(main.cuda_stream))} |"
    for s in lens:
        Y = lambda st: ops.unique_phase(s, st)
        def conc(first):
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            if first == "X": X(side.cuda_stream); Y(main.cuda_stream)
            else: Y(main.cuda_stream); X(side.cuda_stream)
            ev2 = torch.cuda.Event(); ev2.record(side); main.wait_event(ev2)
        line += f" S={s}: Y {t(lambda: Y(main.cuda_stream))} seq {t(lambda: (X(main.cuda_stream), Y(main.cuda_stream)))} X||Y {t(lambda: conc('X'))} Y||X {t(lambda: conc('Y'))} |"
    print(line, flush=True)
