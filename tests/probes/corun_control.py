import ctypes as C, sys, torch
sys.path.insert(0, ".")
dev = "cuda:0"
lib = C.CDLL("build_probe/libcorun.so")
for f in (lib.corun_launch, lib.corun_launch_small): f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(16, device=dev); out2 = torch.zeros(16, device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream(); side2 = torch.cuda.Stream()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main); fn(); e1.record(main); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return torch.tensor(ts).mean().item()
for name, f in (("small", lib.corun_launch_small), ("big", lib.corun_launch)):
    one = lambda: f(out.data_ptr(), 800, 1, main.cuda_stream)
    two_seq = lambda: (f(out.data_ptr(), 800, 1, main.cuda_stream), f(out2.data_ptr(), 800, 1, main.cuda_stream))
    def two_par(a=side, b=main):
        ev = torch.cuda.Event(); ev.record(main); a.wait_event(ev)
        if b is not main: b.wait_event(ev)
        f(out.data_ptr(), 800, 1, a.cuda_stream); f(out2.data_ptr(), 800, 1, b.cuda_stream)
        ev2 = torch.cuda.Event(); ev2.record(a); main.wait_event(ev2)
        if b is not main:
            ev3 = torch.cuda.Event(); ev3.record(b); main.wait_event(ev3)
    print(f"{name}: one {t(one):6.1f}  two in order {t(two_seq):6.1f}  two on (side, main) {t(two_par):6.1f}  two on (side, side2) {t(lambda: two_par(side, side2)):6.1f}", flush=True)
