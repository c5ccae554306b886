#!/usr/bin/env python3
"""One-call form vs two-stream form of the decode operator, decided once (development tool, GPU box):

    python tools/two_stream_ab.py [--iters 30] [--model]

Per shape (C2, its tensor-parallel shards, the C5 slice, C3), a HIP graph of `hydragen_attention_nopad` in each form:
  (ii)  the reference's protocol: replays timed one by one, 512 MB write flush + 512 MB read before each
        (/root/reference/hydragen/benchmark_utils.py:140-170, scripts/microbenchmark.py:24-47)
  (iii) replays back to back without a sync in between (what a decode loop of 32 layers does): us per replay
and with --model (i): `generate()` of a random-weight Llama-2-7B at B = 1024, P = 2048, ms per decode step in both modes."""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hydragen_amd import attention
from hydragen_amd.attention import hydragen_attention_nopad

DEV = "cuda:0"
SHAPES = [
    ("C2 (B=1024,P=2048,32/32)", 1024, 2048, (32, 64, 128), 32, 32),
    ("C2 TP=2 shard (16/16)", 1024, 2048, (64,), 16, 16),
    ("C2 TP=4 shard (8/8)", 1024, 2048, (64,), 8, 8),
    ("C2 TP=8 shard (4/4)", 1024, 2048, (64, 128), 4, 4),
    ("C5 TP=8 slice (B=2048,P=4096,8/1)", 2048, 4096, (128, 256), 8, 1),
    ("C3 (B=64,P=16384,32/8)", 64, 16384, (256,), 32, 8),
]


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def one_by_one(g, iters, flush, clean):
    for _ in range(3):
        g.replay()
    ts = []
    for _ in range(iters):
        flush.add_(1)
        clean.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = torch.tensor(ts)
    return t.mean().item(), t.std().item()


def back_to_back(g, iters):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--model", action="store_true")
    a = ap.parse_args()
    flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)
    clean = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(0)
    print("| shape | S | one-call flushed (ii) | two-stream flushed (ii) | one-call back to back (iii) | two-stream back to back (iii) |")
    print("|---|---|---|---|---|---|")
    for name, B, P, Ss, Hq, Hkv in SHAPES:
        r = lambda *s: torch.randn(*s, device=DEV, dtype=torch.bfloat16, generator=gen)  # noqa: E731
        q, kv, sk, sv = r(B, 1, Hq, 128), r(2, B, max(Ss), Hkv, 128), r(1, P, Hkv, 128), r(1, P, Hkv, 128)
        for S in Ss:
            lens = torch.full((B,), S, dtype=torch.int32, device=DEV)
            cells = {}
            for mode in ("off", "on"):
                prev = attention.set_two_stream(mode)
                try:
                    g = capture(lambda: hydragen_attention_nopad(q, kv[0], kv[1], [sk], [sv], seq_len=lens))
                finally:
                    attention.set_two_stream(prev)
                cells[mode] = (one_by_one(g, a.iters, flush, clean), back_to_back(g, 4 * a.iters))
                del g
            print(f"| {name} | {S} | {cells['off'][0][0]:7.1f} ± {cells['off'][0][1]:4.1f} | {cells['on'][0][0]:7.1f} ± {cells['on'][0][1]:4.1f} | "
                  f"{cells['off'][1]:7.1f} | {cells['on'][1]:7.1f} |", flush=True)
        del q, kv, sk, sv
    del flush, clean
    torch.cuda.empty_cache()
    if a.model:
        from hydragen_amd.llama import HydragenLlamaForCausalLM, LlamaConfig

        cfg = LlamaConfig.llama2_7b()
        model = HydragenLlamaForCausalLM.from_config(cfg, dtype=torch.bfloat16, device=DEV, seed=0)
        model.graph(True)
        prompt = torch.randint(1, cfg.vocab_size, (1, 2048), device=DEV)

        def run(new):
            model.setup_caches(max_unique_batch_size=1024, max_unique_seq_length=128 + 16, max_shared_batch_sizes=[1], max_shared_seq_lengths=[2048])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(input_ids=prompt, num_return_sequences=1024, max_new_tokens=new, temperature=100.0)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        print("\n| generate() Llama-2-7B B=1024 P=2048, 128 new tokens (i) | ms per decode step |")
        print("|---|---|")
        for mode in ("off", "auto", "off", "auto"):
            prev = attention.set_two_stream(mode)
            try:
                model.graph(False)
                model.graph(True)  # drop the captured decode graph: the form is baked into it
                run(4)
                full = min(run(128) for _ in range(2))
                pre = min(run(1) for _ in range(2))
            finally:
                attention.set_two_stream(prev)
            print(f"| two-stream {mode} | {(full - pre) / 127 * 1e3:7.3f} |", flush=True)


if __name__ == "__main__":
    main()
