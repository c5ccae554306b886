"""Do the decode GEMMs of Llama-2-7B at batch 1024 run faster from a [K, N] (pre-transposed) weight than from the
nn.Linear [N, K] layout?  Each shape cycles through 8 weight tensors (cold weights, as in the model's layer sequence)."""
import torch
dev = "cuda:0"
M = 1024
shapes = {"qkv": (4096, 12288), "o": (4096, 4096), "gate_up": (4096, 22016), "down": (11008, 4096)}


def timeit(fs, n=6):
    for f in fs:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for f in fs:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / len(fs) * 1e3


for name, (K, N) in shapes.items():
    x = torch.randn(M, K, device=dev).bfloat16()
    Ws = [(0.02 * torch.randn(N, K, device=dev)).bfloat16() for _ in range(8)]
    Wts = [w.t().contiguous() for w in Ws]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tn = timeit([lambda w=w: torch.nn.functional.linear(x, w) for w in Ws])
    nn_ = timeit([lambda w=w: torch.matmul(x, w) for w in Wts])
    tn_out = timeit([lambda w=w: torch.mm(x, w.t(), out=out) for w in Ws])
    fl = 2.0 * M * K * N
    print(f"{name:8s} [N,K] linear {tn:7.1f} us ({fl / tn / 1e6:6.0f} TF)   [K,N] matmul {nn_:7.1f} us ({fl / nn_ / 1e6:6.0f} TF)   [N,K] mm(out=) {tn_out:7.1f} us")
