#!/usr/bin/env python3
"""
bench.py -- decode attention hot path (Hydragen decomposed shared-prefix attention) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): bf16, batch 1024, one shared prefix of 2048 tokens, Llama-2-7B head
config (32 q heads = 32 kv heads, head_dim 128), unique suffix anywhere in 1..128 tokens.

One "step" = ONE pass of the hot path for one decode token of one layer at ONE fixed suffix length s:
`hydragen_attention` (prefix pass + suffix pass + fused LSE combine) over the whole batch.  The K timed steps
sample s uniformly over 1..128 whatever K is (`suffix_schedule`: s_i = 1 + floor(frac((i + 1/2) * c / K) * 128),
c = ceil(K / 128)), so the mean suffix is 64.5 for K = 20 as for K = 256; the lengths actually run are printed in
`config`.  value = batch * steps / wall time (tokens/s through the attention layer), all inputs resident in HBM
before the timed region.  Each step issues the product's `hyd_decode_attn_fused` as its two documented phases
(HYD_PHASE_SHARED, HYD_PHASE_UNIQUE: the same two kernel launches as the one-call form) with a HIP event between
them, so each kernel's duration is measured inside the timed region on the launch stream.

N > 1: tensor-parallel head sharding exactly as /root/reference/hydragen/tp.py:90-124 (Hq/N query heads and
Hkv/N kv heads per rank, batch replicated) with the per-layer all-reduce(sum) of the [B, 1, hidden] attention
block output (tp.py:108-112) on RCCL, timed with its own events.  Total work is fixed -> "strong" scaling.

Extra objects on the JSON line (rank 0):
  roofline            the kernel that took most of the timed region (by its events): algorithmic bytes or flops /
                      mean launch duration against the gfx950 peak; the other kernel in `roofline_other`
  reference_protocol  the reference's own timing protocol (hydragen/benchmark_utils.py:82-170,
                      scripts/microbenchmark.py:24-47): operator captured in a HIP graph, replays timed one by one
                      with a 512 MB cache flush in between, mean / std / rstd per sweep point s = 16..128
                      (docs/sweeps_from_paper.md:159-161), next to the no-sharing baseline under the same protocol
  accuracy            measured bf16 error of the operator at this shape against fp64 attention over [prefix; suffix]
  model_decode        decode tokens/s of a random-weight Llama-2-7B through HydragenLlamaForCausalLM.generate
                      (scripts/synth.py:33-79,207-226 protocol), N = 1 only
  cpu_baseline        oracle/cpu_port_torch.py (a port: the reference has no CPU path) on the host cores
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import threading
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# (round 6: no "achievable" constant any more -- round 5's 6300 GB/s flattered the line, the same run streamed 6.8 TB/s elsewhere.
#  roofline.achievable_peak is the best HBM rate MEASURED IN THIS RUN by an HBM-bound launch of this library: the no-sharing
#  leg's suffix pass over [prefix + suffix] private keys per sequence, or the timed steps' own best suffix length.)
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak
SWEEP = (16, 32, 48, 64, 80, 96, 112, 128)  # docs/sweeps_from_paper.md:159-161 restricted to C2's 0..128
# BASELINE.json configs that bench.py runs as a workload (the others are parity-test cases: tests/test_fullsize_gpu.py)
WORKLOADS = {
    "c2": dict(batch=1024, prefix=2048, max_suffix=128, qheads=32, kvheads=32,
               name="C2 decode attention layer-step (Llama-2-7B head config)"),
    "c5": dict(batch=2048, prefix=4096, max_suffix=256, qheads=64, kvheads=8,
               name="C5 decode attention layer-step (Llama-3-70B head config, whole job: 64 q / 8 kv heads)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="BASELINE.json config: c2 (the metric's: batch 1024, prefix 2048, 32/32 heads, suffix 1..128) or c5 "
                         "(Llama-3-70B head config whole-job: batch 2048, prefix 4096, 64 q / 8 kv heads, suffix 1..256; "
                         "sharded over heads at N > 1 as tp.py:90-124); --batch/--prefix/... override single fields")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--prefix", type=int, default=None)
    ap.add_argument("--max-suffix", type=int, default=None)
    ap.add_argument("--qheads", type=int, default=None)
    ap.add_argument("--kvheads", type=int, default=None)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--region-timeout", type=float, default=300.0,
                    help="N > 1: seconds a rank waits inside the warm-up / timed region before it reports an error line and exits "
                         "(a peer that never arrives must not hang the job)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nosharing", action="store_true")
    ap.add_argument("--no-protocol", action="store_true", help="skip the graph + flush reference-protocol sweep")
    ap.add_argument("--no-model", action="store_true", help="skip the Llama-2-7B decode tokens/s leg")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--no-paper-sweep", action="store_true", help="skip the reference microbenchmark's own sweep corners (8 q / 1 kv heads)")
    ap.add_argument("--step-form", choices=("eager", "graph"), default="eager",
                    help="how a timed step without events issues the operator: one eager C call (hyd_decode_attn_fused; the default since round 6: "
                         "2-4 us per step less than a graph replay on this stack, tests/probes/eager_vs_graph_probe.py) or one replay of its "
                         "captured HIP graph (rounds 1-5; the reference's own protocol, hydragen/benchmark_utils.py:140-170, stays graph-based in "
                         "`reference_protocol`); the other form is timed once in `trials.other_form_us_per_step`")
    ap.add_argument("--two-stream", action="store_true",
                    help="steps replay the two-stream form of the operator instead of the one-call form (A/B; DESIGN 4.8)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not count HBM bytes with rocprofv3 --pmc child passes of this command (then: the committed passes, if they match)")
    ap.add_argument("--kv-candidates", type=int, default=None,
                    help="candidate placements tried for the unique K|V arena before anything is timed (hydragen_amd/placement.py; "
                         "default: the package's, 6; 1 = plain allocation)")
    ap.add_argument("--trials", type=int, default=3, help="repetitions of the K-step schedule after the headline region (spread)")
    ap.add_argument("--protocol-iters", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--model-new-tokens", type=int, default=128)
    ap.add_argument("--detail-out", default="", help="where the full (uncompacted) result goes; default gpurun_out/bench_detail.json (scratch)")
    ap.add_argument("--no-xgmi", action="store_true", help="N > 1: skip the direct xGMI all-reduce (hyd_allreduce_sum) leg")
    ap.add_argument("--no-graph-collective", action="store_true",
                    help="N > 1: skip the HIP-graph capture of (attention + all-reduce), llama.py:849-854")
    return apply_workload(ap.parse_args())


def apply_workload(args):
    """Fill the shape fields a command line left unset from its --workload preset."""
    w = WORKLOADS[args.workload]
    for field, key in (("batch", "batch"), ("prefix", "prefix"), ("max_suffix", "max_suffix"), ("qheads", "qheads"), ("kvheads", "kvheads")):
        if getattr(args, field) is None:
            setattr(args, field, w[key])
    return args


def suffix_schedule(steps: int, smax: int) -> list[int]:
    """Suffix length of every timed step: a uniform cover of 1..smax for ANY step count (>= 8 to be meaningful)."""
    c = max(1, math.ceil(steps / smax))
    return [1 + int((((i + 0.5) * c / steps) % 1.0) * smax) for i in range(steps)]


def launch_schedule(steps: int, warmup: int, trials: int, smax: int) -> list[int]:
    """Suffix length of EVERY launch of the operator in a run with the untimed legs switched off (what a profiler sees):
    3 eager warm-ups + 1 capture per distinct length (`_capture`; a capture records, it does not launch), the warm-up
    steps, the timed steps, the repeats."""
    sched, warm = suffix_schedule(steps, smax), suffix_schedule(max(warmup, 1), smax)[:warmup]
    return [s for s in sorted(set(sched + suffix_schedule(max(warmup, 1), smax))) for _ in range(3)] + warm + sched * (1 + trials)


def describe_schedule(sched: list[int]) -> str:
    if len(sched) <= 32:
        return ",".join(map(str, sched))
    return f"{len(sched)} steps, min {min(sched)}, max {max(sched)}, mean {sum(sched) / len(sched):.2f} (uniform cover)"


class Ops:
    """Pre-marshalled `hyd_decode_params` per suffix length, and per suffix length one captured HIP graph of the
    operator in its two-stream form (shared phase on a side stream || unique phase, then the merge) -- what
    `hydragen_attention` issues while the decode loop's graph is captured (hydragen_amd/attention.py::_launch_decode)."""

    def __init__(self, q, k, v, sk, sv, lens_needed, two_stream=True):
        from hydragen_amd import _lib, attention
        from hydragen_amd._lib import (DecodeParams, HYD_PHASE_ALL, HYD_PHASE_MERGE, HYD_PHASE_SHARED, HYD_PHASE_UNIQUE,
                                       HYD_PHASE_UNIQUE_PARTIAL)
        from hydragen_amd.attention import _fill_level
        from hydragen_amd.flash import fill_suffix_params

        self.lib = _lib.load()
        self._lib = _lib
        self.two_stream = two_stream
        self.side = torch.cuda.Stream()
        B = q.shape[0]
        self.out = torch.empty_like(q)
        self.params, self.keep, self.graphs = {}, [], {}
        ws_bytes = 0
        for s in sorted(set(lens_needed)):
            sl = torch.full((B,), s, dtype=torch.int32, device=q.device)
            ps = {}
            for phase in (HYD_PHASE_ALL, HYD_PHASE_SHARED, HYD_PHASE_UNIQUE, HYD_PHASE_UNIQUE_PARTIAL, HYD_PHASE_MERGE):
                p = DecodeParams()
                fill_suffix_params(p.suffix, q, k, v, sl, self.out)
                p.n_levels = 1
                p.phase = phase
                _fill_level(p.levels[0], sk, sv, None, None, False, B)
                ps[phase] = p
            # the two-stream form's shared phase keeps to half of the chip (persistent prefix workgroups)
            ps["shared_side"] = DecodeParams.from_buffer_copy(ps[HYD_PHASE_SHARED])
            ps["shared_side"].shared_max_workgroups = attention.TWO_STREAM_PREFIX_CUS
            ps["f32_partials"] = DecodeParams.from_buffer_copy(ps[HYD_PHASE_ALL])
            ps["f32_partials"].f32_partials = 1
            ws_bytes = max(ws_bytes, self.lib.hyd_decode_workspace_bytes(C.byref(ps["f32_partials"])))
            ws_bytes = max(ws_bytes, self.lib.hyd_decode_workspace_bytes(C.byref(ps[HYD_PHASE_ALL])))
            self.params[s] = ps
            self.keep.append(sl)
        self.ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=q.device)
        for ps in self.params.values():
            for p in ps.values():
                p.workspace, p.workspace_bytes = self.ws.data_ptr(), ws_bytes
        self.PH = (HYD_PHASE_ALL, HYD_PHASE_SHARED, HYD_PHASE_UNIQUE, HYD_PHASE_UNIQUE_PARTIAL, HYD_PHASE_MERGE)

    def _call(self, s, which, stream):
        self._lib.check(self.lib.hyd_decode_attn_fused(C.byref(self.params[s][which]), stream))

    def fused(self, s, stream):
        self._call(s, self.PH[0], stream)

    def shared_phase(self, s, stream):
        self._call(s, self.PH[1], stream)

    def unique_phase(self, s, stream):
        self._call(s, self.PH[2], stream)

    def two_stream_issue(self, s):
        """fork: shared phase on the side stream || unique partial on the current stream; join; merge"""
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        self._call(s, "shared_side", self.side.cuda_stream)
        self._call(s, self.PH[3], main.cuda_stream)
        main.wait_stream(self.side)
        self._call(s, self.PH[4], main.cuda_stream)

    def graph(self, s, two_stream=None):
        """The operator at suffix length s as a captured HIP graph (hydragen/benchmark_utils.py:140-170 captures the
        operator the same way); two-stream form unless told otherwise."""
        two = self.two_stream if two_stream is None else two_stream
        key = (s, two)
        if key not in self.graphs:
            fn = (lambda: self.two_stream_issue(s)) if two else (lambda: self.fused(s, torch.cuda.current_stream().cuda_stream))
            self.graphs[key] = _capture(fn)
        return self.graphs[key]

    def step(self, s, form="graph"):
        """One pass of the operator: a replay of its captured graph, or (form "eager", one-call form only) one C call on the current stream."""
        if form == "eager" and not self.two_stream:
            self.fused(s, torch.cuda.current_stream().cuda_stream)
        else:
            self.graph(s).replay()


def _respawn(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-execute this command line under torch.distributed.run, one
    rank per GPU on this node (the bootstrap of /root/reference/hydragen/utils.py:118-133 reads the same environment)."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_respawn(args.gpus))  # plain `python bench.py --gpus N`: start the N ranks ourselves
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU; there is no CPU fallback for the product path"
    # Self-test knobs (not used by the driver): HYD_BENCH_BACKEND=gloo + HYD_BENCH_ONE_DEVICE=1 run the
    # N > 1 control flow with every rank on cuda:0 (RCCL refuses two ranks on one device).
    backend = os.environ.get("HYD_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("HYD_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pre = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        pre = preflight(args, rank, world, backend, dev)  # initialises the process group (stage rccl_init); exits 3 on a hard failure

    B, P, S, D = args.batch, args.prefix, args.max_suffix, args.dim
    assert args.qheads % world == 0 and args.kvheads % world == 0, "heads must divide the TP degree (tp.py:43-46)"
    Hq, Hkv = args.qheads // world, args.kvheads // world  # tp.py:103-106,121-123
    hidden = args.qheads * D
    dt = torch.bfloat16
    torch.manual_seed(1234 + rank)
    q = torch.randn(B, 1, Hq, D, device=dev, dtype=dt)
    # one arena, K | V, placed as PerLayerKVCache places a layer's unique caches (hydragen_amd/placement.py: candidates timed with
    # the suffix pass before anything else is allocated or measured; the report goes on the result line)
    from hydragen_amd import placement
    if args.kv_candidates is not None:
        placement.set_candidates(args.kv_candidates)
    (kv,), kv_place = placement.place_kv_arenas(1, (B, S, Hkv, D), dt, dev, Hq, zero=False)
    kv.normal_()
    k, v = kv[0], kv[1]
    sk = torch.randn(1, P, Hkv, D, device=dev, dtype=dt)
    sv = torch.randn(1, P, Hkv, D, device=dev, dtype=dt)
    sched = suffix_schedule(args.steps, S)
    warm_sched = suffix_schedule(max(args.warmup, 1), S)
    sweep = [s for s in SWEEP + (160, 192, 224, 256) if s <= S]
    ops = Ops(q, k, v, sk, sv, sched + warm_sched + sweep, two_stream=args.two_stream)
    for s_ in sorted(set(sched + warm_sched)):  # capture outside the timed region
        ops.graph(s_)
    # stand-in for the row-parallel o_proj partial output that tp.py:108-112 all-reduces
    ar_buf = torch.randn(B, 1, hidden, device=dev, dtype=dt) if world > 1 else None
    ar_host = ar_buf.float().cpu() if (world > 1 and backend != "nccl") else None
    stream = torch.cuda.current_stream().cuda_stream

    def collective():
        if backend == "nccl":
            dist.all_reduce(ar_buf)
        else:  # gloo self-test: reduce a host copy
            dist.all_reduce(ar_host)

    def step(s, ev=None, form=None):
        if ev is None:  # the operator in one piece: one eager C call (two kernel launches) or one replay of its captured graph
            ops.step(s, form or args.step_form)
            if world > 1:
                collective()
            return
        # steps that carry events: the two kernels in order, eagerly, an event before, between and after them
        ev[0].record()
        ops.shared_phase(s, stream)
        ev[1].record()
        ops.unique_phase(s, stream)
        ev[2].record()
        if world > 1:
            collective()
            ev[3].record()

    # N > 1: a rank that never arrives (a peer died, a collective that never completes) must end the job with an error
    # line, not hang it: every rank arms a watchdog around the warm-up, the timed region and the repeats.
    region_dog = _region_watchdog(args, rank, world) if world > 1 else None
    for i in range(args.warmup):
        step(warm_sched[i])
    torch.cuda.synchronize()

    # Per-kernel durations need the two kernels one after the other with an event between them, so the steps that carry
    # events run the operator eagerly IN ORDER (three records cost ~10 us of such a step); all other steps replay the
    # captured graph.  Event steps: i % 5 == 1 in the first half of the schedule and their mirror images K-1-i (one step
    # in five; K = 20: suffix 10, 42, 87, 119) -- the schedule is a uniform cover, so this subset has the mean suffix length of
    # all K steps.  (Until round 5: i % 4 == 1, six of twenty steps; the event records are instrumentation that the timed region
    # pays for, ~10 us per such step, and timing events inside a captured graph are refused on this stack:
    # tests/probes/graph_event_probe.py.)  Fewer than 8 steps: every step.
    K_ = args.steps
    ev_idx = sorted({i for i in range(K_ // 2) if i % 5 == 1} | {K_ - 1 - i for i in range(K_ // 2) if i % 5 == 1}) \
        if K_ >= 8 else list(range(K_))
    nev = 4 if world > 1 else 3
    ev_of = {i: [torch.cuda.Event(enable_timing=True) for _ in range(nev)] for i in ev_idx}
    events = [ev_of[i] for i in ev_idx]
    for ev in events:  # materialise the hipEvent_t handles (torch creates them lazily)
        for e_ in ev:
            e_.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(sched[i], ev_of.get(i))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- spread: the same K-step schedule again, `--trials` times, without events (the headline stays trial 0 above) ----
    trial_us = []
    for _ in range(args.trials):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(sched[i])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_ = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt_], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        trial_us.append(dt_ / args.steps * 1e6)

    other_form = "graph" if (args.step_form == "eager" and not args.two_stream) else "eager"
    other_us = None
    if args.trials and not args.two_stream:  # the same schedule once more in the OTHER step form (graph replays <-> eager calls)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(sched[i], form=other_form)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        other_us = (time.perf_counter() - t1) / args.steps * 1e6

    if region_dog is not None:
        region_dog.cancel()

    # ---- per-kernel durations from the HIP events recorded inside the timed region ----------
    e = 2
    pre_ms = [ev[0].elapsed_time(ev[1]) for ev in events]
    suf_ms = [ev[1].elapsed_time(ev[2]) for ev in events]
    n_ev = len(ev_idx)
    sched_ev = [sched[i] for i in ev_idx]
    suf_bytes_of = lambda s: 2 * e * Hkv * D * B * s + 2 * B * Hq * D * e + 4 * B * Hq
    suf_bytes = [suf_bytes_of(s) for s in sched_ev]
    pre_flops = 4.0 * B * Hq * P * D
    suf_gbs = sum(suf_bytes) / (sum(suf_ms) * 1e-3) / 1e9
    pre_tflops = pre_flops * n_ev / (sum(pre_ms) * 1e-3) / 1e12
    suffix_roof = {
        "kernel": ("suffix_attn_gqa_kernel (matrix-core suffix pass for grouped-query heads + fused LSE combine)" if args.qheads // args.kvheads >= 4
                   else "suffix_attn_rows_kernel (token-row suffix pass + fused LSE combine)" if Hq == Hkv and Hkv % 4 == 0
                   else "suffix_attn_kernel (suffix pass + fused LSE combine)"),
        "bound": "hbm", "achieved": suf_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": suf_gbs / HBM_PEAK_GBS, "traffic": None,
        "avg_launch_us": sum(suf_ms) / n_ev * 1e3,
        "algorithmic_bytes_per_launch_mean": sum(suf_bytes_of(s) for s in sched) / args.steps,
        "share_of_timed_region": sum(suf_ms) / (sum(suf_ms) + sum(pre_ms)),
    }
    # the best rate an HBM-bound launch of this library reached IN THIS RUN (so far: the timed steps' best suffix length; the
    # no-sharing leg below replaces it when it streams faster) -- MI355X_MICROARCH.md: 8.0 TB/s is the spec
    best_step = max(suf_bytes[i] / (suf_ms[i] * 1e-3) / 1e9 for i in range(n_ev))
    suffix_roof.update(achievable_peak=best_step, achievable_peak_source="best timed step of this run (by its events)",
                       frac_of_achievable=suf_gbs / best_step)
    prefix_roof = {
        "kernel": "prefix_attn kernel (batched-query MFMA pass over the shared prefix)",
        "bound": "mfma", "achieved": pre_tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": pre_tflops / MFMA_PEAK_TFLOPS, "traffic": None,
        "avg_launch_us": sum(pre_ms) / n_ev * 1e3,
        "flops_per_launch": pre_flops,
        "share_of_timed_region": sum(pre_ms) / (sum(suf_ms) + sum(pre_ms)),
    }
    _attach_traffic(suffix_roof, prefix_roof, args, world)
    dominant_is_suffix = sum(suf_ms) >= sum(pre_ms)

    # achieved fraction per suffix bucket (the fixed cost of the suffix pass shows at small s)
    buckets = {}
    for lo, hi in ((1, 16), (17, 32), (33, 64), (65, 96), (97, 128)):
        idx = [i for i, s in enumerate(sched_ev) if lo <= s <= hi]
        if idx:
            g = sum(suf_bytes[i] for i in idx) / (sum(suf_ms[i] for i in idx) * 1e-3) / 1e9
            buckets[f"{lo}-{hi}"] = {"steps": len(idx), "GB/s": g, "frac": g / HBM_PEAK_GBS,
                                     "suffix_us": sum(suf_ms[i] for i in idx) / len(idx) * 1e3}

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
            "workload": f"{WORKLOADS[args.workload]['name']}: batch {B}, shared prefix {P}, {args.qheads}q/{args.kvheads}kv heads "
                        f"d={D}, one hydragen_attention call per step at a fixed suffix length; suffix lengths of the timed "
                        f"steps: {describe_schedule(sched)}",
            "preset": args.workload, "batch": B, "prefix_len": P, "suffix_lens": sched if len(sched) <= 64 else describe_schedule(sched),
            "suffix_len_mean": sum(sched) / len(sched), "qheads": args.qheads, "kvheads": args.kvheads, "head_dim": D,
            "parallelism": f"tp{world} (heads sharded, all-reduce [B,1,{hidden}] bf16 per step)" if world > 1 else "single GPU",
        },
        # `value` is an attention-LAYER rate; the whole-model decode rate is the next key (filled by the model leg, N = 1, c2)
        "value_is": "batch / one layer's attention time (attn_us_per_step); NOT decode throughput: that is decode_tokens_per_sec",
        "decode_tokens_per_sec": None,
        "attn_us_per_step": elapsed / args.steps * 1e6,
        "prefix_us": sum(pre_ms) / n_ev * 1e3,
        "suffix_us_mean": sum(suf_ms) / n_ev * 1e3,
        # where the unique K|V arena sits in HBM was chosen by timing candidates (us of one suffix pass over all keys each), before the timed region
        "kv_placement": ({"candidates": kv_place["candidates"], "probe_us": kv_place["probe_us"], "kept": kv_place["kept"][0],
                          "spacer_gib": round(kv_place["spacer_bytes"] / 2**30, 1)} if kv_place.get("probed") else
                         {"candidates": 1, "why": kv_place.get("why")}),
        "step_forms": {"form": "graph" if args.two_stream else args.step_form, "steps_in_that_form": args.steps - n_ev,
                       "graph_form": "one-call form (hyd_decode_attn_fused: prefix pass, then suffix pass with the merge in its epilogue)"
                       if not args.two_stream else
                       "two-stream form: shared phase (persistent prefix workgroups on half of the CUs) on a side stream || "
                       "unique phase, join, log-sum-exp merge (hydragen_amd.attention.set_two_stream)",
                       "eager_in_order_steps_with_events": n_ev},
        "events": {"steps_with_events": n_ev, "rule": "i % 5 == 1 in the first half of the schedule + mirror images K-1-i" if n_ev < args.steps else "every step",
                   "steps": ev_idx, "suffix_lens": sched_ev, "suffix_len_mean": sum(sched_ev) / n_ev,
                   "why": "a kernel's own duration needs the two kernels in order with an event between them; these steps run the "
                          "operator eagerly in that form, the per-kernel rooflines are theirs"},
        "trials": {"headline": "trial 0 = the timed region above (HIP events on one step in five)", "trial0_us_per_step": elapsed / args.steps * 1e6,
                   "repeat_us_per_step": trial_us, "repeat_note": "same schedule, every step in the headline's step form, no events, each bracketed like the headline",
                   "other_form": other_form, "other_form_us_per_step": other_us,
                   **({"repeat_mean_us": sum(trial_us) / len(trial_us), "repeat_min_us": min(trial_us), "repeat_max_us": max(trial_us)} if trial_us else {})},
        "suffix_frac_by_suffix_len": buckets,
        "roofline": suffix_roof if dominant_is_suffix else prefix_roof,
        "roofline_other": prefix_roof if dominant_is_suffix else suffix_roof,
    }
    if world > 1:
        res["preflight"] = pre
        ar_ms = [ev[2].elapsed_time(ev[3]) for ev in events]
        res["allreduce_us"] = sum(ar_ms) / n_ev * 1e3
        res["allreduce_bytes"] = ar_buf.numel() * 2
        res["rccl_ranks"] = dist.get_world_size()
        res["collective_backend"] = backend
        if not args.no_xgmi:
            res["allreduce_xgmi"] = _guarded(lambda: xgmi_allreduce_leg(ar_buf), 120.0, res, rank, key="allreduce_xgmi")
        if backend == "nccl" and not args.no_graph_collective:
            res["graph_collective"] = _guarded(lambda: graph_collective(ops, sweep[len(sweep) // 2], ar_buf), 120.0,
                                               res, rank)

    solo = rank == 0 and world == 1
    if solo and not args.no_protocol:
        res["reference_protocol"] = reference_protocol(ops, q, sk, sv, k, v, sweep, args.protocol_iters,
                                                       with_nosharing=not args.no_nosharing)
        ns = res["reference_protocol"].get("nosharing_speedup_mean")
        if ns is not None:
            res["nosharing_speedup"] = ns
        nsr = res["reference_protocol"].get("nosharing_GBs_at_max_suffix")
        if nsr is not None and nsr > suffix_roof["achievable_peak"]:
            suffix_roof.update(achievable_peak=nsr, frac_of_achievable=suffix_roof["achieved"] / nsr,
                               achievable_peak_source="this run's no-sharing leg: the same library's suffix pass over prefix + suffix private keys per sequence")
    if rank == 0 and not args.no_accuracy:
        res["accuracy"] = accuracy(ops, q, sk, sv, k, v, S // 2)
    if solo and not args.no_paper_sweep:
        ops = q = k = v = sk = sv = None  # 2.3 GB of C2 tensors and the captured graphs: free them for the sweep's 18 GB
        torch.cuda.empty_cache()
        res["paper_sweep"] = _guarded(lambda: paper_sweep(max(4, args.protocol_iters // 3)), 120.0, res, rank, key="paper_sweep")
    if solo and not args.no_model and args.workload == "c2":  # the model leg is Llama-2-7B at C2's batch and prefix
        ops = None
        torch.cuda.empty_cache()
        try:
            res["model_decode"] = model_decode(B, P, args.model_new_tokens)
            res["decode_tokens_per_sec"] = res["model_decode"]["decode_tokens_per_s"]
        except Exception as ex:  # the attention line above must survive a failure of the model leg
            res["model_decode"] = {"error": f"{type(ex).__name__}: {ex}"}
    if rank == 0 and not args.no_cpu_baseline:  # whole-job head counts, so that it compares with `value` at any N
        res["cpu_baseline"] = cpu_baseline(B, P, args.qheads, args.kvheads, D, args.cpu_seconds)
    if rank == 0:
        if args.detail_out:
            global DETAIL_FILE
            DETAIL_FILE = Path(args.detail_out).resolve()
        print(json.dumps(compact_line(res)))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _setup_caches_one_aliased_unique_buffer(model, max_unique_batch_size, max_unique_seq_length, max_shared_batch_sizes, max_shared_seq_lengths):
    """What `setup_caches` does (hydragen_amd/llama.py; /root/reference/hydragen/llama.py:921-955), except that every layer's
    unique K/V cache is ONE pair of buffers.  A timing stand-in for the no-sharing baseline at the headline batch
    (scripts/synth.py:112,151 replicates the prefix into every sequence's cache: 36 GB per LAYER at batch 1024 / prefix 2048):
    the generated tokens are meaningless -- the layers overwrite each other's keys -- the time per step is not, every layer
    still appends to and streams the whole buffer from HBM.  Lives here, not on the model's API, which mirrors the reference's."""
    from hydragen_amd.llama import PerLayerKVCache

    model.maybe_invalidate()
    max_unique_seq_length = (max_unique_seq_length + 15) // 16 * 16
    first = None
    for layer in model.model.layers:
        alias = first is not None
        cache = PerLayerKVCache(
            max_unique_batch_size=1 if alias else max_unique_batch_size, max_unique_seq_length=16 if alias else max_unique_seq_length,
            max_shared_batch_sizes=max_shared_batch_sizes, max_shared_seq_lengths=max_shared_seq_lengths,
            n_kv_heads=model.config.num_key_value_heads, head_dim=model.config.hidden_size // model.get_num_heads(),
            device=model.lm_head.weight.device, dtype=model.lm_head.weight.dtype)
        if alias:
            cache.per_completion_k_cache = first.per_completion_k_cache
            cache.per_completion_v_cache = first.per_completion_v_cache
        else:
            first = cache
        layer.self_attn.kv_cache = cache
    model.kv_cache_allocated = True


DETAIL_FILE = REPO / "gpurun_out" / "bench_detail.json"  # scratch (untracked); --detail-out moves it; tools/profile_round.sh copies the round's into profiles/
LINE_BUDGET = 7600  # bytes: the driver's record keeps the last 8 KB of the line


def _sig(x, n=5):
    return float(f"{x:.{n}g}") if isinstance(x, float) and math.isfinite(x) else x


def _round_floats(o):
    if isinstance(o, dict):
        return {k: _round_floats(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_round_floats(v) for v in o]
    return _sig(o)


def compact_line(res: dict) -> dict:
    """The ONE line the driver records must fit its 8 KB tail.  The full result (per-point std / rstd / n of the reference
    protocol, the paper sweep's rows as objects, the prose that says how each figure was taken) goes to
    the detail file (DETAIL_FILE / --detail-out); the line keeps every figure's MEAN, five significant digits, and short notes."""
    try:
        DETAIL_FILE.parent.mkdir(parents=True, exist_ok=True)
        DETAIL_FILE.write_text(json.dumps(res, indent=1))
        detail = str(DETAIL_FILE.relative_to(REPO)) if DETAIL_FILE.is_relative_to(REPO) else str(DETAIL_FILE)
    except OSError as ex:  # a read-only tree must not cost the line
        detail = f"not written ({type(ex).__name__})"
    keep = ("value", "ms_per_step", "attn_us_per_step")  # the contract's own figures stay as measured
    line = {k: (v if k in keep else _round_floats(v)) for k, v in res.items()}
    line["detail_file"] = detail
    rp = line.get("reference_protocol")
    if isinstance(rp, dict) and "by_suffix_len" in rp:
        cols = sorted({c for v in rp["by_suffix_len"].values() for c in v if isinstance(v[c], dict)})
        rp["by_suffix_len"] = {"columns": cols + ["speedup"], "mean_us": {s_: [v.get(c, {}).get("mean_us") for c in cols] + [v.get("speedup")]
                                                                      for s_, v in rp["by_suffix_len"].items()}}
        rp["protocol"] = "graph replays timed one by one, 512 MB write flush between them (benchmark_utils.py:82-170); per-point std / rstd / n in detail_file"
    ps = line.get("paper_sweep")
    if isinstance(ps, dict) and "rows" in ps:
        cols = ["batch", "prefix", "suffix", "hydragen_us", "nosharing_us", "speedup"]
        ps["rows"] = {"columns": cols, "values": [[r_.get(c) for c in cols] for r_ in ps["rows"]]}
    for roof in ("roofline", "roofline_other"):
        src = line.get(roof, {}).get("traffic_source")
        if isinstance(src, str) and len(src) > 110:
            line[roof]["traffic_source"] = src[:107] + "..."
    for path in (("events", "why"), ("cpu_baseline", "sample"), ("step_forms", "graph_form"), ("trials", "repeat_note"), ("trials", "headline")):
        d = line.get(path[0])
        if isinstance(d, dict) and isinstance(d.get(path[1]), str) and len(d[path[1]]) > 100:
            d[path[1]] = d[path[1]][:97] + "..."
    if isinstance(line.get("events"), dict):
        line["events"].pop("steps", None)
    acc = line.get("accuracy")
    if isinstance(acc, dict):  # the top level IS the one-call form; keep the other forms' headline figure only
        acc.pop("one_call_form", None)
        for form in ("two_stream_form", "one_call_form_fp32_prefix_partial"):
            if isinstance(acc.get(form), dict):
                acc[form] = {k: acc[form][k] for k in ("relative_l2", "us_back_to_back") if k in acc[form]}
    cfg = line.get("config")
    if isinstance(cfg, dict) and isinstance(cfg.get("workload"), str) and "; suffix lengths of the timed steps:" in cfg["workload"]:
        cfg["workload"] = cfg["workload"].split("; suffix lengths of the timed steps:")[0] + " (suffix lengths of the timed steps: config.suffix_lens)"
    # last resort, in order of dispensability
    for drop in (("paper_sweep", "rows"), ("reference_protocol", "by_suffix_len"), ("suffix_frac_by_suffix_len",), ("accuracy",), ("events",)):
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        if len(drop) == 1:
            line[drop[0]] = "see detail_file"
        elif isinstance(line.get(drop[0]), dict):
            line[drop[0]][drop[1]] = "see detail_file"
    return line


def _live_traffic(args):
    """HBM bytes per launch of the two kernels, COUNTED NOW: two `rocprofv3 --pmc` child passes (FETCH_SIZE, WRITE_SIZE;
    counters only, no trace domain) of this very command line with the untimed legs off, read from the rocpd database.
    FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 bytes, MI355X_MICROARCH.md, HBM section); both are KB.
    None when rocprofv3 is missing, when this process is itself being profiled, or when a pass fails."""
    import shutil, sqlite3, subprocess, tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if Path("/opt/rocm/bin/rocprofv3").exists() else None)
    if exe is None or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    child = [sys.executable, str(REPO / "bench.py"), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch),
             "--prefix", str(args.prefix), "--max-suffix", str(args.max_suffix), "--qheads", str(args.qheads), "--kvheads", str(args.kvheads),
             "--dim", str(args.dim), "--trials", "0", "--no-cpu-baseline", "--no-protocol", "--no-model", "--no-accuracy",
             "--no-paper-sweep", "--no-nosharing", "--no-live-traffic", "--kv-candidates", "1"]  # (bytes per launch do not depend on the placement; its probe launches would)
    env = dict(os.environ, TMPDIR="/tmp")
    means = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                subprocess.run([exe, "--pmc", counter, "-d", d, "-o", "run", "--"] + child, cwd="/tmp", env=env, timeout=240,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
                dbs = list(Path(d).rglob("*.db"))
                if not dbs:
                    return None
                cur = sqlite3.connect(str(dbs[0])).cursor()
                for name, n, avg in cur.execute("select kernel_name, count(*), avg(value) from counters_collection "
                                                "where counter_name = ? group by kernel_name", (counter,)):
                    key = "suffix" if "suffix_attn" in name else "prefix" if "prefix_attn" in name else None
                    if key:
                        means[(key, counter)] = (float(avg), int(n))
    except Exception:  # a profiler that cannot run here must never cost the bench line
        return None
    launches = launch_schedule(args.steps, args.warmup, 0, args.max_suffix)
    e = 2
    out = {"launches": len(launches),
           "suffix_algorithmic": sum(2 * e * args.kvheads * args.dim * args.batch * s_ + 2 * args.batch * args.qheads * args.dim * e
                                     + 4 * args.batch * args.qheads for s_ in launches) / len(launches),
           "prefix_algorithmic": 2 * args.prefix * args.kvheads * args.dim * e + 2 * args.batch * args.qheads * args.dim * e + 4 * args.batch * args.qheads}
    for key in ("suffix", "prefix"):
        f, w = means.get((key, "FETCH_SIZE")), means.get((key, "WRITE_SIZE"))
        if f is None or w is None or f[1] != len(launches) or w[1] != len(launches):
            return None  # not the launch schedule this accounting assumes
        out[key] = (2.0 * f[0] + w[0]) * 1024.0
    return out


def _attach_traffic(suffix_roof, prefix_roof, args, world):
    """roofline.traffic = HBM bytes per launch.  First choice: counted in this run by rocprofv3 --pmc child passes of the
    same command line (_live_traffic).  Otherwise the committed passes (profiles/traffic_latest.json), and then only when
    they were collected with THIS command line (same shape, same suffix schedule); else just their traffic / algorithmic
    ratio."""
    if world != 1:  # the counters describe the single-GPU shape
        return
    live = None if args.no_live_traffic else _live_traffic(args)
    if live is not None:
        for roof, key in ((suffix_roof, "suffix"), (prefix_roof, "prefix")):
            roof["traffic"] = live[key]
            roof["traffic_over_algorithmic"] = live[key] / live[f"{key}_algorithmic"]
            roof["traffic_source"] = (f"counted in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child passes, counters "
                                      f"only) of this command line with the untimed legs off; mean over the {live['launches']} launches of a pass "
                                      "(capture warm-ups, warm-up steps, timed steps: bench.launch_schedule); FETCH_SIZE doubled per the "
                                      "gfx950 note in MI355X_MICROARCH.md")
        return
    tr = REPO / "profiles" / "traffic_latest.json"
    if not tr.exists():
        return
    t = json.loads(tr.read_text())
    same = (t.get("steps") == args.steps and t.get("batch") == args.batch and t.get("prefix") == args.prefix
            and t.get("max_suffix") == args.max_suffix and t.get("qheads") == args.qheads and t.get("kvheads") == args.kvheads)
    for roof, key in ((suffix_roof, "suffix"), (prefix_roof, "prefix")):
        b = t.get(f"{key}_hbm_bytes_per_launch")
        if b is None:
            continue
        roof["traffic_source"] = t.get("source")
        if same:
            roof["traffic"] = b
        if t.get(f"{key}_algorithmic_bytes_per_launch"):
            roof["traffic_over_algorithmic"] = b / t[f"{key}_algorithmic_bytes_per_launch"]


PREFLIGHT_STAGE_SECONDS = 60.0


def preflight(args, rank, world, backend, dev):
    """N > 1, before anything is timed: the pieces a first run on a multi-GPU node can fail in, one by one, each under its own
    60 s watchdog (180 s for the process group's bootstrap), so that a failure NAMES its stage instead of hanging the job or surfacing as a wrong number later
    (the bootstrap and collectives of /root/reference/hydragen/utils.py:118-133, tp.py:108-112, the in-graph collective of
    llama.py:849-854).  Rank 0 prints one `[preflight] {json}` line per stage.  Stages:
      devices          visible devices >= N (one-device self-test mode excepted)
      rccl_init        init_process_group with N ranks + a 1 KiB all-reduce checked against the expected sum
      peer_access      hipDeviceCanAccessPeer for every ordered pair of the N devices
      xgmi_allreduce   one hyd_allreduce_sum (XgmiAllReduce) of 1 MiB checked against the process group's result  [--no-xgmi skips]
      graph_collective a captured all-reduce replayed twice, result checked                                       [--no-graph-collective skips]
    A failed or hung devices / rccl_init / peer_access stage ends the job: error line naming the stage, exit code 3.  A FAILED
    xgmi_allreduce or graph_collective stage only switches that optional leg off -- the RCCL headline still runs; a HUNG one
    cannot be recovered from (the process group is stuck) and exits 3, the line says which flag skips the stage.
    Returns {stage: {...}} for the result line; may set args.no_xgmi / args.no_graph_collective."""
    report, one_device = {}, bool(os.environ.get("HYD_BENCH_ONE_DEVICE"))

    def say(stage, rec):
        report[stage] = rec
        if rank == 0:
            print("[preflight] " + json.dumps({"stage": stage, **rec}))
            sys.stdout.flush()

    def die(stage, why, hint=""):
        line = {"metric": "decode_attention_tokens_per_sec", "value": None, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "error": f"preflight stage '{stage}' failed on rank {rank}: {why}" + (f" ({hint})" if hint else ""),
                "preflight_stage": stage, "rank": rank, "preflight": report}
        print(json.dumps(line))
        sys.stdout.flush()
        os._exit(3)

    def staged(stage, fn, optional_flag=None, seconds=PREFLIGHT_STAGE_SECONDS):
        """Run one stage under the watchdog.  Hard stages die on any failure; optional ones only on a hang."""
        hint = f"--{optional_flag.replace('_', '-')} skips this stage" if optional_flag else ""
        timer = threading.Timer(seconds, lambda: die(stage, f"no answer within {seconds:.0f} s", hint))
        timer.daemon = True
        timer.start()
        t0 = time.perf_counter()
        try:
            if os.environ.get("HYD_BENCH_FAIL_STAGE") == stage:  # self-test knob (tests/test_bench.py), like HYD_BENCH_BACKEND
                raise RuntimeError("failure injected by HYD_BENCH_FAIL_STAGE")
            rec = fn() or {}
            say(stage, {"ok": True, "seconds": round(time.perf_counter() - t0, 3), **rec})
            return True
        except Exception as ex:  # noqa: BLE001 -- every failure must name its stage
            why = f"{type(ex).__name__}: {ex}"
            if optional_flag is None:
                timer.cancel()
                die(stage, why)
            say(stage, {"ok": False, "seconds": round(time.perf_counter() - t0, 3), "error": why[:300], "consequence": f"leg switched off ({hint})"})
            setattr(args, optional_flag, True)
            return False
        finally:
            timer.cancel()

    def reduce_any(t):
        """all-reduce(sum) through the process group: RCCL on the device, or (gloo self-test) a host copy."""
        if backend == "nccl":
            dist.all_reduce(t)
            return t
        h = t.float().cpu()
        dist.all_reduce(h)
        return h.to(device=t.device, dtype=t.dtype)

    def st_devices():
        n = torch.cuda.device_count()
        if not one_device and n < world:
            raise RuntimeError(f"{n} visible devices for {world} ranks")
        return {"visible_devices": n, "one_device_self_test": one_device}

    def st_init():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        x = torch.full((256,), float(rank + 1), device=dev, dtype=torch.float32)  # 1 KiB
        got = reduce_any(x)
        torch.cuda.synchronize()
        want = world * (world + 1) / 2
        if not bool((got == want).all()):
            raise RuntimeError(f"1 KiB all-reduce returned {float(got[0])}, expected {want}")
        return {"backend": backend, "ranks": dist.get_world_size(), "all_reduce_1KiB": "sum checked"}

    def st_peer():
        n = world if not one_device else 1
        m = [[int(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]
        if not all(all(r) for r in m):
            raise RuntimeError(f"hipDeviceCanAccessPeer matrix has holes: {m}")
        return {"can_access_peer": m}

    def st_xgmi():
        from hydragen_amd.xgmi_allreduce import XgmiAllReduce

        comm = XgmiAllReduce(max_bytes=1 << 20, timeout_log2_polls=24)  # ~5 s per wait: well inside the stage's watchdog
        try:
            gen = torch.Generator(device=dev).manual_seed(77 + rank)
            x = torch.randn(1 << 19, device=dev, dtype=torch.bfloat16, generator=gen)  # 1 MiB
            want = reduce_any(x.clone())
            got = comm.all_reduce_(x.clone())
            torch.cuda.synchronize()
            err = float((got.float() - want.float()).abs().max())
            st = comm.status()
            # same addends, another order of summation in bf16: a few units in the last place of O(sqrt(world)) values
            if not (err <= 0.0625 * max(1.0, world ** 0.5)) or not torch.isfinite(got.float()).all():
                raise RuntimeError(f"1 MiB hyd_allreduce_sum differs from the process group's result by {err} (status {st})")
            dist.barrier()
            return {"bytes": 1 << 20, "max_abs_diff_vs_group": err, "uncached_block": bool(comm.uncached), "status": st}
        finally:
            comm.close()

    def st_graph():
        if backend != "nccl":
            return {"skipped": "the in-graph collective is RCCL's (llama.py:849-854); this run uses " + backend}
        x = torch.full((1 << 18,), float(rank + 1), device=dev, dtype=torch.float32)  # 1 MiB
        y = torch.empty_like(x)

        def fn():
            y.copy_(x)
            dist.all_reduce(y)

        g = _capture(fn)
        want = world * (world + 1) / 2
        for i in range(2):
            y.zero_()
            g.replay()
            torch.cuda.synchronize()
            if not bool((y == want).all()):
                raise RuntimeError(f"replay {i} of a captured all-reduce returned {float(y[0])}, expected {want}")
        return {"replays": 2, "bytes": 1 << 20}

    def agree(flag):
        """An optional leg runs on every rank or on none: one rank's failed stage switches it off everywhere."""
        off = reduce_any(torch.tensor([1.0 if getattr(args, flag) else 0.0], device=dev))
        torch.cuda.synchronize()
        if float(off[0]) > 0:
            setattr(args, flag, True)

    staged("devices", st_devices)
    staged("rccl_init", st_init, seconds=3 * PREFLIGHT_STAGE_SECONDS)  # (the first RCCL bootstrap of a cold node detects the topology)
    staged("peer_access", st_peer)
    if not args.no_xgmi:
        staged("xgmi_allreduce", st_xgmi, "no_xgmi")
    if not args.no_graph_collective:
        staged("graph_collective", st_graph, "no_graph_collective")
    staged("agreement", lambda: (agree("no_xgmi"), agree("no_graph_collective"),
                                 {"xgmi_leg": not args.no_xgmi, "graph_collective_leg": not args.no_graph_collective})[2])
    return report


def _region_watchdog(args, rank, world):
    """Timer that ends this rank with an error line when the warm-up / timed region of an N > 1 run does not finish in
    --region-timeout seconds.  Every rank prints the line (a hung rank 0 cannot speak for the others); the exit code is
    non-zero so that the launcher tears the other ranks down."""
    def bail():
        line = {"metric": "decode_attention_tokens_per_sec", "value": None, "unit": "tokens/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "error": f"rank {rank}: the timed region did not finish within "
                f"{args.region_timeout:.0f} s (a peer never arrived or a collective never completed)", "rank": rank}
        print(json.dumps(line))
        sys.stdout.flush()
        os._exit(3)

    timer = threading.Timer(args.region_timeout, bail)
    timer.daemon = True
    timer.start()
    return timer


def _guarded(fn, seconds, res, rank, key="graph_collective"):
    """Run an optional leg; if it hangs (a collective that never completes), print the line without it and leave."""
    def bail():
        res[key] = {"error": f"timed out after {seconds:.0f} s"}
        if rank == 0:
            print(json.dumps(res))
            sys.stdout.flush()
        os._exit(0)

    timer = threading.Timer(seconds, bail)
    timer.daemon = True
    timer.start()
    try:
        return fn()
    except Exception as ex:
        return {"error": f"{type(ex).__name__}: {ex}"}
    finally:
        timer.cancel()


def _capture(fn):
    """HIP-graph capture after eager warm-ups on a side stream (hydragen/benchmark_utils.py:140-170)."""
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


def _stats(us):
    t = torch.tensor(us, dtype=torch.float64)
    mean, std = float(t.mean()), float(t.std()) if len(us) > 1 else 0.0
    return {"mean_us": mean, "std_us": std, "rstd": std / mean if mean else 0.0, "n": len(us)}


def _timed_replays(graph, iters, flush, clean=None):
    for _ in range(3):
        graph.replay()
    out = []
    for _ in range(iters):
        if flush is not None:
            flush.add_(1)  # 512 MB read-modify-write: evicts L2 and the 256 MB Infinity Cache (microbenchmark.py:24-47)
        if clean is not None:
            clean.sum()    # 512 MB read only: pushes the flush's DIRTY lines out before the timed replay
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return out


def xgmi_allreduce_leg(ar_buf):
    """N > 1: the same [B, 1, hidden] tensor through the library's two-shot direct all-reduce over IPC-mapped peer
    blocks (hyd_allreduce_sum) -- result checked against RCCL's, then timed with HIP events."""
    from hydragen_amd.xgmi_allreduce import XgmiAllReduce

    comm = XgmiAllReduce(max_bytes=ar_buf.numel() * ar_buf.element_size())
    x = torch.randn_like(ar_buf)
    if dist.get_backend() == "nccl":
        want = x.clone()
        dist.all_reduce(want)
    else:  # gloo self-test: reduce a host copy
        h = x.float().cpu()
        dist.all_reduce(h)
        want = h.to(x.device)
    got = comm.all_reduce_(x.clone())
    torch.cuda.synchronize()
    err = float((got.float() - want.float()).abs().max())
    for _ in range(5):
        comm.all_reduce_(x)
    torch.cuda.synchronize()
    dist.barrier()
    us = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        comm.all_reduce_(x)
        e1.record()
        torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
    st = comm.status()
    dist.barrier()
    comm.close()
    return {**_stats(us), "bytes": ar_buf.numel() * ar_buf.element_size(), "max_abs_diff_vs_rccl": err, "status": st,
            "what": "hyd_allreduce_sum: stage + two-shot direct exchange, eager, back to back"}


def graph_collective(ops, s, ar_buf):
    """N > 1: the decode step's attention + its all-reduce inside one HIP graph (llama.py:849-854 captures the
    whole TP forward, NCCL all-reduce included)."""
    def fn():
        ops.fused(s, torch.cuda.current_stream().cuda_stream)
        dist.all_reduce(ar_buf)

    g = _capture(fn)
    us = _timed_replays(g, 20, None)
    return {"suffix_len": s, **_stats(us), "what": "graph replay of hyd_decode_attn_fused + RCCL all-reduce, back to back"}


def reference_protocol(ops, q, sk, sv, k, v, sweep, iters, with_nosharing):
    from hydragen_amd.flash import flash_attention_seqlen

    dev = q.device
    B, _, Hq, D = q.shape
    P = sk.shape[1]
    flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=dev)
    clean = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=dev)
    rows = {}
    for s in sweep:
        g = ops.graph(s)
        rows[s] = {"hydragen_flushed": _stats(_timed_replays(g, iters, flush)),
                   "hydragen_flushed_clean": _stats(_timed_replays(g, iters, flush, clean)),
                   "hydragen_back_to_back": _stats(_timed_replays(g, iters, None)),
                   "two_stream_flushed_clean": _stats(_timed_replays(ops.graph(s, two_stream=True), iters, flush, clean)),
                   "two_stream_back_to_back": _stats(_timed_replays(ops.graph(s, two_stream=True), iters, None))}
    del clean
    out = {
        "protocol": "HIP-graph replay of the operator (one-call form unless --two-stream; two_stream_* = the two-stream form: it wins when every replay is "
                    "timed alone and loses its gain to the two cross-queue edges when replays follow each other, DESIGN 4.8), HIP events per replay, 512 MB flush between "
                    "replays (and the same replays back to back); hydragen/benchmark_utils.py:82-170, scripts/microbenchmark.py:24-47. "
                    "The flush is a WRITE, as the reference's: on MI355X the 256 MB Infinity Cache then holds dirty lines "
                    "whose write-back is charged to the timed call (C5: +45 us); hydragen_flushed_clean reads a second "
                    "512 MB buffer after the flush, so the timed call starts cold but with nothing to write back",
        "iters": iters, "by_suffix_len": rows,
    }
    flagged = [s for s, r in rows.items() if r["hydragen_flushed"]["rstd"] > 0.10]  # scripts/synth.py:240-245
    if flagged:
        out["rstd_over_10pct_at"] = flagged
    out["hydragen_flushed_mean_us"] = sum(r["hydragen_flushed"]["mean_us"] for r in rows.values()) / len(rows)
    out["hydragen_flushed_clean_mean_us"] = sum(r["hydragen_flushed_clean"]["mean_us"] for r in rows.values()) / len(rows)
    out["hydragen_back_to_back_mean_us"] = sum(r["hydragen_back_to_back"]["mean_us"] for r in rows.values()) / len(rows)
    out["two_stream_back_to_back_mean_us"] = sum(r["two_stream_back_to_back"]["mean_us"] for r in rows.values()) / len(rows)
    out["two_stream_flushed_clean_mean_us"] = sum(r["two_stream_flushed_clean"]["mean_us"] for r in rows.values()) / len(rows)
    if not with_nosharing:
        return out
    # no-sharing FlashAttention-equivalent (scripts/microbenchmark.py:91-127 go_baseline with --unique-seq-len): every
    # sequence owns a private [P + S] KV; same HIP suffix kernel with seq_len = P + s
    S = k.shape[1]
    try:
        kt = torch.empty(B, P + S, sk.shape[2], D, device=dev, dtype=q.dtype)
        vt = torch.empty_like(kt)
    except torch.OutOfMemoryError:
        out["nosharing_error"] = "not enough HBM for the materialised no-sharing KV"
        return out
    kt[:, :P] = sk
    vt[:, :P] = sv
    kt[:, P:] = k
    vt[:, P:] = v
    sp = []
    for s in sweep:
        sl = torch.full((B,), P + s, dtype=torch.int32, device=dev)
        g = _capture(lambda: flash_attention_seqlen(q, kt, vt, seq_len=sl))
        st = _stats(_timed_replays(g, max(4, iters // 4), flush))
        del g
        rows[s]["nosharing_flushed"] = st
        rows[s]["speedup"] = st["mean_us"] / rows[s]["hydragen_flushed"]["mean_us"]
        sp.append(rows[s]["speedup"])
    byts = 2 * 2 * B * (P + sweep[-1]) * sk.shape[2] * D
    out["nosharing_GBs_at_max_suffix"] = byts / (rows[sweep[-1]]["nosharing_flushed"]["mean_us"] * 1e-6) / 1e9
    out["nosharing_speedup_mean"] = sum(sp) / len(sp)
    out["nosharing_speedup_min"] = min(sp)
    return out


def paper_sweep(iters):
    """The reference's own attention microbenchmark (scripts/microbenchmark.py, docs/sweeps_from_paper.md:152-169): its default
    heads (8 query / 1 kv, d=128, microbenchmark.py:136-138), bf16, (batch, prefix) of the paper's sweep, suffix 0 / 128 / 512,
    operator captured in a HIP graph, 512 MB flush between timed replays; Hydragen vs the no-sharing baseline."""
    from hydragen_amd.attention import hydragen_attention_nopad
    from hydragen_amd.flash import flash_attention_seqlen

    dev, dt, Hq, Hkv, D = "cuda:0", torch.bfloat16, 8, 1, 128
    flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.int32, device=dev)
    rows = []
    for B, P in ((512, 1024), (1024, 2048), (2048, 4096), (4096, 8192)):
        q = torch.randn(B, 1, Hq, D, device=dev, dtype=dt)
        sk, sv = torch.randn(1, P, Hkv, D, device=dev, dtype=dt), torch.randn(1, P, Hkv, D, device=dev, dtype=dt)
        k, v = torch.randn(B, 512, Hkv, D, device=dev, dtype=dt), torch.randn(B, 512, Hkv, D, device=dev, dtype=dt)
        kt = torch.cat([sk.expand(B, -1, -1, -1), k], 1).contiguous()
        vt = torch.cat([sv.expand(B, -1, -1, -1), v], 1).contiguous()
        for S in (0, 128, 512):
            lens = torch.full((B,), S, dtype=torch.int32, device=dev)
            tl = lens + P
            if S == 0:  # microbenchmark.py:76-83: no unique keys -> the prefix-only early exit (attention.py:273-274)
                e = torch.empty(B, 0, Hkv, D, device=dev, dtype=dt)
                gh = _capture(lambda: hydragen_attention_nopad(q, e, e, [sk], [sv]))
            else:
                ks, vs = k[:, :S], v[:, :S]
                gh = _capture(lambda: hydragen_attention_nopad(q, ks, vs, [sk], [sv], seq_len=lens))
            gn = _capture(lambda: flash_attention_seqlen(q, kt, vt, seq_len=tl))
            h, n = _stats(_timed_replays(gh, iters, flush)), _stats(_timed_replays(gn, max(3, iters // 2), flush))
            rows.append({"batch": B, "prefix": P, "suffix": S, "hydragen_us": h["mean_us"], "hydragen_rstd": h["rstd"],
                         "nosharing_us": n["mean_us"], "speedup": n["mean_us"] / h["mean_us"]})
            del gh, gn
        del kt, vt, k, v
        torch.cuda.empty_cache()
    return {"heads": "8 q / 1 kv, d=128 (scripts/microbenchmark.py:136-138)", "iters": iters,
            "protocol": "HIP-graph replay, 512 MB write flush before every timed replay (scripts/microbenchmark.py:24-47)", "rows": rows}


def accuracy(ops, q, sk, sv, k, v, s):
    """Measured error of the bf16 operator at this shape, in both forms: 32 sequences x all heads against fp64 softmax
    attention over the concatenated [prefix; suffix] keys of the same bf16 inputs (torch, on the GPU)."""
    B, _, Hq, D = q.shape
    g = Hq // sk.shape[2]
    idx = torch.linspace(0, B - 1, 32, device=q.device).long()
    qs = q[idx].double()                                          # [n,1,Hq,D]
    kk = torch.cat([sk.expand(len(idx), -1, -1, -1), k[idx, :s]], 1).double().repeat_interleave(g, 2)
    vv = torch.cat([sv.expand(len(idx), -1, -1, -1), v[idx, :s]], 1).double().repeat_interleave(g, 2)
    sc = torch.einsum("bqhd,bkhd->bhqk", qs, kk) / math.sqrt(D)
    want = torch.einsum("bhqk,bkhd->bqhd", sc.softmax(-1), vv)

    def measure():
        torch.cuda.synchronize()
        got = ops.out[idx].double()
        diff = (got - want).abs()
        return {"max_abs_err": float(diff.max()), "mean_abs_err": float(diff.mean()),
                "relative_l2": float((got - want).norm() / want.norm()),
                "mean_rdiff": float((2 * diff / (got.abs() + want.abs() + 1e-8)).mean())}   # hydragen/utils.py:13-15

    ops.out.zero_()
    ops.two_stream_issue(s)
    two = measure()
    ops.out.zero_()
    ops.fused(s, torch.cuda.current_stream().cuda_stream)
    one = measure()
    # the same with the prefix partial kept fp32 (hyd_decode_params.f32_partials): one rounding less, 16 MiB more traffic
    st = torch.cuda.current_stream().cuda_stream
    ops.out.zero_()
    ops._call(s, "f32_partials", st)
    f32p = measure()
    f32p["us_back_to_back"] = _time_calls(lambda: ops._call(s, "f32_partials", st))
    one["us_back_to_back"] = _time_calls(lambda: ops.fused(s, st))
    return {
        "dtype": "bf16", "suffix_len": s, "sequences": int(len(idx)),
        "reference": "fp64 softmax attention over [prefix; suffix] on the same bf16 inputs",
        **(two if ops.two_stream else one),
        "one_call_form": one, "two_stream_form": two, "one_call_form_fp32_prefix_partial": f32p,
        "bf16_half_ulp_of_max_output": float(want.abs().max()) * 2.0 ** -9,
    }


def _time_calls(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def model_decode(B, P, new_tokens):
    """scripts/synth.py protocol: generate(num_return_sequences=B, max_new_tokens=S, temperature=100) from a P-token
    prompt on a random-weight Llama-2-7B, HIP-graph decode; prefill isolated by a max_new_tokens=1 run (:207-226)."""
    from hydragen_amd.llama import HydragenLlamaForCausalLM, LlamaConfig

    cfg = LlamaConfig.llama2_7b()
    cfg.max_position_embeddings = max(cfg.max_position_embeddings, P + new_tokens + 16)
    dev = "cuda:0"
    model = HydragenLlamaForCausalLM.from_config(cfg, dtype=torch.bfloat16, device=dev, seed=0)
    model.graph(True)
    prompt = torch.randint(1, cfg.vocab_size, (1, P), device=dev)
    model.setup_caches(max_unique_batch_size=B, max_unique_seq_length=new_tokens + 16, max_shared_batch_sizes=[1],
                       max_shared_seq_lengths=[P])

    def run(n, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(input_ids=prompt, num_return_sequences=B, max_new_tokens=n, temperature=100.0, **kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def mode(**kw):
        run(4, **kw)  # warm-up incl. graph capture
        full = min(run(new_tokens, **kw) for _ in range(2))
        pre = min(run(1, **kw) for _ in range(2))
        return full, pre, full - pre

    def small_batch_modes(Bs=128, new=64):
        """hydragen vs hydragen_noshared (scripts/synth.py:112,151: the prefix KV replicated into every sequence's unique
        cache, llama.py:264-298) at the largest batch whose replicated prefix fits in HBM comfortably (1.1 GB per sequence)."""
        out = {"batch": Bs, "new_tokens": new}
        for name, kw, extra in (("hydragen", {}, 0), ("hydragen_noshared", {"disable_hydragen": True}, P)):
            model.setup_caches(max_unique_batch_size=Bs, max_unique_seq_length=new + 16 + extra, max_shared_batch_sizes=[1],
                               max_shared_seq_lengths=[P])

            def go(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.generate(input_ids=prompt, num_return_sequences=Bs, max_new_tokens=n, temperature=100.0, **kw)
                torch.cuda.synchronize()
                return time.perf_counter() - t0

            go(4)
            d = go(new) - go(1)
            out[name] = {"decode_s": d, "decode_tokens_per_s": Bs * (new - 1) / d, "ms_per_decode_step": d / (new - 1) * 1e3}
        out["speedup_vs_noshared"] = out["hydragen"]["decode_tokens_per_s"] / out["hydragen_noshared"]["decode_tokens_per_s"]
        return out

    def noshared_full_batch(new=6):
        """hydragen_noshared AT THE HEADLINE BATCH: one private [B, P + new] K/V buffer (36 GB at B = 1024, P = 2048) aliased
        by all layers (_setup_caches_one_aliased_unique_buffer): every layer streams it from HBM in every step,
        which is what the mode costs; 32 private copies (1.2 TB) would not fit."""
        _setup_caches_one_aliased_unique_buffer(model, B, new + 16 + P, [1], [P])

        def go(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(input_ids=prompt, num_return_sequences=B, max_new_tokens=n, temperature=100.0, disable_hydragen=True)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        go(3)
        d = min(go(new) for _ in range(2)) - min(go(1) for _ in range(2))
        cache = model.model.layers[0].self_attn.kv_cache.per_completion_k_cache
        return {"batch": B, "new_tokens": new, "decode_s": d, "decode_tokens_per_s": B * (new - 1) / d,
                "ms_per_decode_step": d / (new - 1) * 1e3, "private_kv_bytes_all_layers_alias": 2 * cache.numel() * cache.element_size(),
                "note": "all layers alias one private K/V buffer (timing stand-in: tokens are meaningless, HBM traffic per step is the mode's)"}

    full, pre, dec = mode()
    # upper bound of scripts/synth.py:111-115 ("noattention": attention replaced by identity on q, llama.py:433-437)
    _, _, dec_na = mode(disable_attention=True)
    steps = new_tokens - 1
    layers = cfg.num_hidden_layers
    return {
        "model": "Llama-2-7B architecture, random weights, bf16, HIP-graph decode", "batch": B, "prefix": P,
        "new_tokens": new_tokens, "total_s": full, "prefill_s": pre, "decode_s": dec,
        "decode_tokens_per_s": B * steps / dec, "ms_per_decode_step": dec / steps * 1e3,
        "noattention": {"decode_s": dec_na, "decode_tokens_per_s": B * steps / dec_na, "ms_per_decode_step": dec_na / steps * 1e3},
        "fraction_of_noattention_bound": dec_na / dec,
        "attention_us_per_layer_step": (dec - dec_na) / steps / layers * 1e6,
        "small_batch_modes": small_batch_modes(),
        f"noshared_b{B}": (lambda r: dict(r, speedup_of_hydragen=(B * steps / dec) / r["decode_tokens_per_s"]))(noshared_full_batch()),
        "protocol": "scripts/synth.py:33-79,111-115,148-178,207-226 (modes hydragen, noattention; hydragen_noshared at the batch that fits)",
    }


def cpu_baseline(B, P, Hq, Hkv, D, budget_s):
    """oracle/cpu_port_torch.py (README.md:377-461 restated) on the host cores; bounded sample."""
    from oracle import cpu_port_torch as port

    logical = os.cpu_count() or 1
    cores = max(1, logical // 2) if logical > 16 else logical  # physical cores (SMT siblings only add contention)
    torch.set_num_threads(cores)
    S = 64
    # the decomposed form fits the host (2 GiB of fp32 unique K/V at C2) and is timed on the WHOLE batch;
    # the time budget bounds the number of iterations, not the workload
    bs = B
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
    c1 = _cpu_c1(port)
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
        "sample": f"decomposed attention (torch CPU fp32, {cores} threads) on all {bs} sequences, all {Hq} heads, "
                  f"prefix {P}, suffix {S} (the mean of the GPU schedule); {n} iterations of the whole step; tokens/s = {bs} / time per step",
        "c1_full": c1,
        "nosharing_tokens_per_sec": bn / t_ns,
        "nosharing_sample": f"no-sharing SDPA over concatenated KV on {bn} sequences (stride-0 expanded prefix), {m} iterations",
        "cpu_model": _cpu_model(),
    }


def _cpu_c1(port):
    """BASELINE config 1 literal (batch 4, prefix 64, suffix 8, 4 heads, dim 64), the reference's CPU-runnable case, in full."""
    g = torch.Generator().manual_seed(1)
    q, k, v = torch.randn(4, 1, 4, 64, generator=g), torch.randn(4, 8, 4, 64, generator=g), torch.randn(4, 8, 4, 64, generator=g)
    sk, sv = torch.randn(1, 64, 4, 64, generator=g), torch.randn(1, 64, 4, 64, generator=g)
    sl = torch.full((4,), 8, dtype=torch.int64)
    nt = torch.get_num_threads()
    torch.set_num_threads(1)  # 70 kFLOP: threads only add wake-up latency
    try:
        for _ in range(20):
            port.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
        n, t0 = 500, time.perf_counter()
        for _ in range(n):
            port.hydragen_attention_nopad(q, k, v, [sk], [sv], sl)
        t = (time.perf_counter() - t0) / n
        for _ in range(20):
            port.nosharing_attention(q, k, v, sk, sv, sl)
        t1 = time.perf_counter()
        for _ in range(n):
            port.nosharing_attention(q, k, v, sk, sv, sl)
        tn = (time.perf_counter() - t1) / n
    finally:
        torch.set_num_threads(nt)
    return {"config": "batch 4, prefix 64, suffix 8, 4/4 heads, d=64, fp32, 1 thread, 500 iterations", "decomposed_us": t * 1e6,
            "nosharing_us": tn * 1e6, "tokens_per_s": 4 / t}


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
