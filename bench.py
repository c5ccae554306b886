#!/usr/bin/env python3
"""
bench.py -- decode attention hot path (Hydragen decomposed shared-prefix attention) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): bf16, batch 1024, one shared prefix of 2048 tokens,
Llama-2-7B head config (32 q heads = 32 kv heads, head_dim 128), unique suffix growing
1 -> 128 tokens.  One "step" = ONE pass of the hot path for one decode token of one layer:
`hydragen_attention` (prefix pass + suffix pass + fused LSE combine) over the whole batch with
suffix length s = 1 + (step mod 128).  value = batch * steps / wall time (tokens/s through the
attention layer), all inputs resident in HBM before the timed region.

N > 1: tensor-parallel head sharding exactly as /root/reference/hydragen/tp.py:90-124
(Hq/N query heads and Hkv/N kv heads per rank, batch replicated) with the per-layer
all-reduce(sum) of the [B, 1, hidden] attention block output (tp.py:108-112) on RCCL.
Total work is fixed -> "strong" scaling.

Extra objects on the JSON line:
  roofline       dominant kernel (suffix pass, HBM-bound): algorithmic bytes / HIP-event time
  roofline_prefix  prefix-pass MFMA utilisation (4*B*Hq*P*D flops / HIP-event time / 2.5 PF/s)
  nosharing      the no-sharing FlashAttention-equivalent baseline on the same GPU (every sequence
                 owns a private [P+S] KV; same suffix kernel) and the speedup over it
  cpu_baseline   oracle/cpu_port_torch.py (a port: the reference has no CPU path) on the host cores
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--prefix", type=int, default=2048)
    ap.add_argument("--max-suffix", type=int, default=128)
    ap.add_argument("--qheads", type=int, default=32)
    ap.add_argument("--kvheads", type=int, default=32)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nosharing", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


class Ops:
    """One decode step = ONE call of the product's fused entry point (`hyd_decode_attn_fused`, what
    hydragen_amd.attention.hydragen_attention issues), through its measurement twin
    `hyd_decode_attn_fused_timed`, which records a HIP event between the prefix pass and the suffix pass
    so that each kernel can be timed inside the timed region."""

    def __init__(self, q, k, v, sk, sv):
        from hydragen_amd import _lib
        from hydragen_amd._lib import DecodeParams
        from hydragen_amd.attention import _fill_level
        from hydragen_amd.flash import fill_suffix_params

        self.lib = _lib.load()
        self._lib = _lib
        B = q.shape[0]
        self.out = torch.empty_like(q)
        self.params, self.keep = {}, []
        ws_bytes = 0
        for s in range(1, k.shape[1] + 1):
            sl = torch.full((B,), s, dtype=torch.int32, device=q.device)
            p = DecodeParams()
            fill_suffix_params(p.suffix, q, k, v, sl, self.out)
            p.n_levels = 1
            _fill_level(p.levels[0], sk, sv, None, None, False, B)
            ws_bytes = max(ws_bytes, self.lib.hyd_decode_workspace_bytes(C.byref(p)))
            self.params[s] = p
            self.keep.append(sl)
        self.ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=q.device)
        for p in self.params.values():
            p.workspace, p.workspace_bytes = self.ws.data_ptr(), ws_bytes

    def step(self, s, stream, ev_mid=None):
        self._lib.check(self.lib.hyd_decode_attn_fused_timed(C.byref(self.params[s]), stream, ev_mid))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU; there is no CPU fallback for the product path"
    # Self-test knobs (not used by the driver): HYD_BENCH_BACKEND=gloo + HYD_BENCH_ONE_DEVICE=1 run the
    # N > 1 control flow with every rank on cuda:0 (RCCL refuses two ranks on one device).
    backend = os.environ.get("HYD_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("HYD_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    B, P, S, D = args.batch, args.prefix, args.max_suffix, args.dim
    assert args.qheads % world == 0 and args.kvheads % world == 0, "heads must divide the TP degree (tp.py:43-46)"
    Hq, Hkv = args.qheads // world, args.kvheads // world  # tp.py:103-106,121-123
    hidden = args.qheads * D
    dt = torch.bfloat16
    torch.manual_seed(1234 + rank)
    q = torch.randn(B, 1, Hq, D, device=dev, dtype=dt)
    k = torch.randn(B, S, Hkv, D, device=dev, dtype=dt)
    v = torch.randn(B, S, Hkv, D, device=dev, dtype=dt)
    sk = torch.randn(1, P, Hkv, D, device=dev, dtype=dt)
    sv = torch.randn(1, P, Hkv, D, device=dev, dtype=dt)
    ops = Ops(q, k, v, sk, sv)
    # stand-in for the row-parallel o_proj partial output that tp.py:108-112 all-reduces
    ar_buf = torch.randn(B, 1, hidden, device=dev, dtype=dt) if world > 1 else None
    stream = torch.cuda.current_stream().cuda_stream

    def suffix_len(i):
        return 1 + (i % S)

    def step(i, ev=None):
        s = suffix_len(i)
        if ev is not None:
            ev[0].record()
            ops.step(s, stream, ev[1].cuda_event)
            ev[2].record()
        else:
            ops.step(s, stream)
        if world > 1:
            if backend == "nccl":
                dist.all_reduce(ar_buf)
            else:  # gloo self-test: reduce a host copy
                dist.all_reduce(ar_host)

    ar_host = ar_buf.float().cpu() if (world > 1 and backend != "nccl") else None
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    for ev in events:  # materialise the hipEvent_t handles (torch creates them lazily)
        ev[1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations from the HIP events recorded inside the timed region ----------
    e = 2
    pre_ms = [ev[0].elapsed_time(ev[1]) for ev in events]
    suf_ms = [ev[1].elapsed_time(ev[2]) for ev in events]
    suf_bytes = [2 * e * Hkv * D * B * suffix_len(i) + 2 * B * Hq * D * e + 4 * B * Hq for i in range(args.steps)]
    pre_flops = 4.0 * B * Hq * P * D
    suf_gbs = sum(suf_bytes) / (sum(suf_ms) * 1e-3) / 1e9
    pre_tflops = pre_flops * args.steps / (sum(pre_ms) * 1e-3) / 1e12

    res = {
        "metric": "decode_attention_tokens_per_sec",
        "value": B * args.steps / elapsed,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"C2 decode attention layer-step: batch {B}, shared prefix {P}, suffix 1..{S} (cyclic), "
                        f"{args.qheads}q/{args.kvheads}kv heads d={D}, one hydragen_attention call per step",
            "batch": B, "prefix_len": P, "suffix_len": f"1..{S}", "qheads": args.qheads, "kvheads": args.kvheads,
            "head_dim": D, "parallelism": f"tp{world} (heads sharded, all-reduce [B,1,{hidden}] bf16 per step)" if world > 1 else "single GPU",
        },
        "attn_us_per_step": elapsed / args.steps * 1e6,
        "prefix_us": sum(pre_ms) / args.steps * 1e3,
        "suffix_us_mean": sum(suf_ms) / args.steps * 1e3,
        "roofline": {
            "kernel": "suffix_attn_kernel (suffix pass + fused LSE combine)",
            "bound": "hbm", "achieved": suf_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": suf_gbs / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch_mean": sum(suf_bytes) / args.steps,
        },
        "roofline_prefix": {
            "kernel": "prefix_attn_pl_kernel (batched-query MFMA pass, software-pipelined)",
            "bound": "mfma", "achieved": pre_tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": pre_tflops / MFMA_PEAK_TFLOPS, "flops_per_launch": pre_flops,
        },
    }

    tr = REPO / "profiles" / "traffic_latest.json"
    if tr.exists():  # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
        t = json.loads(tr.read_text())
        res["roofline"]["traffic"] = t.get("suffix_hbm_bytes_per_launch")
        res["roofline"]["traffic_source"] = t.get("source")
        res["roofline_prefix"]["traffic"] = t.get("prefix_hbm_bytes_per_launch")
    if rank == 0 and world == 1 and not args.no_nosharing:
        res["nosharing"] = bench_nosharing(q, sk, sv, k, v, S, B * 1.0, res)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(B, P, Hq, Hkv, D, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_nosharing(q, sk, sv, k, v, S, B, res):
    """No-sharing FlashAttention-equivalent decode (scripts/microbenchmark.py:91-127 go_baseline with
    --unique-seq-len): every sequence owns a private [P+S] KV; same suffix kernel, seq_len = P + s."""
    from hydragen_amd.flash import flash_attention_seqlen

    Bi, _, Hq, D = q.shape
    P = sk.shape[1]
    try:
        kt = torch.empty(Bi, P + S, sk.shape[2], D, device=q.device, dtype=q.dtype)
        vt = torch.empty_like(kt)
        kt[:, :P] = sk
        vt[:, :P] = sv
        kt[:, P:] = k
        vt[:, P:] = v
    except torch.OutOfMemoryError:
        return {"error": "not enough HBM for the materialised no-sharing KV"}
    times = {}
    for s in (1, S // 2, S):
        sl = torch.full((Bi,), P + s, dtype=torch.int32, device=q.device)
        for _ in range(2):
            flash_attention_seqlen(q, kt, vt, seq_len=sl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            flash_attention_seqlen(q, kt, vt, seq_len=sl)
        e1.record()
        torch.cuda.synchronize()
        times[s] = e0.elapsed_time(e1) / n * 1e3
    mean_us = sum(times.values()) / len(times)
    byts = 2 * 2 * Bi * (P + S // 2) * sk.shape[2] * D
    return {
        "us_per_step_at_suffix": times,
        "tokens_per_sec": Bi / (mean_us * 1e-6),
        "achieved_GBs_at_mid": byts / (times[S // 2] * 1e-6) / 1e9,
        "hydragen_speedup": mean_us / res["attn_us_per_step"],
        "note": "same HIP suffix kernel over a private [P+S] KV per sequence; seq_len = P + s",
    }


def cpu_baseline(B, P, Hq, Hkv, D, budget_s):
    """oracle/cpu_port_torch.py (README.md:377-461 restated) on the host cores; bounded sample."""
    from oracle import cpu_port_torch as port

    logical = os.cpu_count() or 1
    cores = max(1, logical // 2) if logical > 16 else logical  # physical cores (SMT siblings only add contention)
    torch.set_num_threads(cores)
    S = 64
    # sample: a slice of the batch (all heads, full prefix, mid suffix), sized to the time budget
    bs = 128
    g = torch.Generator().manual_seed(0)
    q = torch.randn(bs, 1, Hq, D, generator=g)
    k = torch.randn(bs, S, Hkv, D, generator=g)
    v = torch.randn(bs, S, Hkv, D, generator=g)
    sk = torch.randn(1, P, Hkv, D, generator=g)
    sv = torch.randn(1, P, Hkv, D, generator=g)
    sl = torch.full((bs,), S, dtype=torch.int64)
    port.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        port.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
        n += 1
        if time.perf_counter() - t0 > budget_s / 2 or n >= 50:
            break
    t_dec = (time.perf_counter() - t0) / n
    # no-sharing form on a smaller slice (it is ~P/S times more work per sequence)
    bn = 16
    t1 = time.perf_counter()
    m = 0
    while True:
        port.nosharing_attention(q[:bn], k[:bn], v[:bn], sk, sv, sl[:bn])
        m += 1
        if time.perf_counter() - t1 > budget_s / 2 or m >= 20:
            break
    t_ns = (time.perf_counter() - t1) / m
    return {
        "value": bs / t_dec, "unit": "tokens/s", "cores": cores, "kind": "port",
        "sample": f"decomposed attention (torch CPU fp32, {cores} threads) on {bs} of {B} sequences, all {Hq} heads, "
                  f"prefix {P}, suffix {S}; {n} iterations; tokens/s = sequences / time per step",
        "nosharing_tokens_per_sec": bn / t_ns,
        "nosharing_sample": f"no-sharing SDPA over concatenated KV on {bn} sequences (stride-0 expanded prefix), {m} iterations",
        "cpu_model": _cpu_model(),
    }


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
