"""The real prefix pass (shared phase) on a plain side stream beside the unique phase on the main stream.
   python tests/probes/corun_real.py"""
import sys, torch
sys.path.insert(0, ".")
import bench
dev = "cuda:0"; dt = torch.bfloat16
B, P, S, H, D = 1024, 2048, 128, 32, 128
q = torch.randn(B, 1, H, D, device=dev, dtype=dt); sk = torch.randn(1, P, H, D, device=dev, dtype=dt); sv = torch.randn_like(sk)
k = torch.randn(B, S, H, D, device=dev, dtype=dt); v = torch.randn_like(k)
lens = [8, 16, 32, 64, 128]
ops = bench.Ops(q, k, v, sk, sv, lens)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
E = lambda: torch.cuda.Event(enable_timing=True)
def seq(s):
    e0, e1 = E(), E(); e0.record(main)
    ops.shared_phase(s, main.cuda_stream); ops.unique_phase(s, main.cuda_stream)
    e1.record(main); torch.cuda.synchronize(); return [e0.elapsed_time(e1) * 1e3]
def par(s, stamps, first):
    e0, xs, xe, ys, ye, e1 = E(), E(), E(), E(), E(), E()
    e0.record(main); side.wait_event(e0)
    def X():
        if stamps: xs.record(side)
        ops.shared_phase(s, side.cuda_stream); xe.record(side)
    def Y():
        if stamps: ys.record(main)
        ops.unique_phase(s, main.cuda_stream)
        if stamps: ye.record(main)
    (X(), Y()) if first == "X" else (Y(), X())
    main.wait_event(xe); e1.record(main); torch.cuda.synchronize()
    r = [e0.elapsed_time(e1) * 1e3]
    if stamps: r += [e0.elapsed_time(t) * 1e3 for t in (xs, xe, ys, ye)]
    return r
def avg(fn, n=8):
    for _ in range(3): fn()
    return torch.tensor([fn() for _ in range(n)]).mean(0).tolist()
for s in lens:
    a = avg(lambda: seq(s))[0]
    b = avg(lambda: par(s, True, "X")); c = avg(lambda: par(s, False, "X"))[0]; d = avg(lambda: par(s, False, "Y"))[0]
    print(f"S={s:4d}: in order {a:6.1f} | side||main with stamps {b[0]:6.1f} (prefix {b[1]:5.1f}..{b[2]:5.1f}, suffix {b[3]:5.1f}..{b[4]:5.1f}) | without stamps {c:6.1f} | suffix launched first {d:6.1f}", flush=True)
