#!/usr/bin/env python3
"""Whole-model decode throughput with random weights (the protocol of /root/reference/scripts/synth.py:
generate(num_return_sequences=B, max_new_tokens=S, temperature=100) from a P-token prompt, modes
hydragen / hydragen_noshared / noattention; prefill isolated by a max_new_tokens=1 run, synth.py:207-226).

    python tools/bench_model.py --model llama2-7b --batch 1024 --prefix 2048 --new 128 --modes hydragen,noattention
Prints one JSON line per mode."""
import argparse, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hydragen_amd.llama import HydragenLlamaForCausalLM, LlamaConfig

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama2-7b")
ap.add_argument("--layers", type=int, default=0, help="override the layer count (0 = architecture's own)")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--prefix", type=int, default=2048)
ap.add_argument("--new", type=int, default=128)
ap.add_argument("--modes", default="hydragen,noattention")
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--tunableop", default="", help="PyTorch TunableOp results file (tools/tune_gemms.py) to select the GEMM solutions from; tuning itself stays off")
ap.add_argument("--tp-slice", type=int, default=1, help="build rank 0's shard of an N-way tensor-parallel model (no collectives: per-GPU compute only)")
a = ap.parse_args()

if a.tunableop:
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(False)
    ok = tun.read_file(a.tunableop)
    print(json.dumps({"tunableop_file": a.tunableop, "read_ok": bool(ok), "enabled": tun.is_enabled(), "entries": len(tun.get_results())}))
cfg = LlamaConfig.llama2_7b() if a.model == "llama2-7b" else LlamaConfig.llama3_70b()
if a.layers:
    cfg.num_hidden_layers = a.layers
cfg.max_position_embeddings = max(cfg.max_position_embeddings, a.prefix + a.new + 16)
dev = "cuda:0"
model = HydragenLlamaForCausalLM.from_config(cfg, dtype=torch.bfloat16, device=dev, seed=0,
                                             tp_shard=(0, a.tp_slice) if a.tp_slice > 1 else None)
model.graph(not a.no_graph)
prompt = torch.randint(1, cfg.vocab_size, (1, a.prefix), device=dev)

def run(mode, new):
    kw = dict(disable_hydragen=(mode == "hydragen_noshared"), disable_attention=(mode == "noattention"))
    uniq = a.new + (a.prefix if mode == "hydragen_noshared" else 0)   # synth.py:56-61
    model.setup_caches(max_unique_batch_size=a.batch, max_unique_seq_length=uniq + 16,
                       max_shared_batch_sizes=[1], max_shared_seq_lengths=[a.prefix])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.generate(input_ids=prompt, num_return_sequences=a.batch, max_new_tokens=new, temperature=100.0, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0

for mode in a.modes.split(","):
    run(mode, 4)  # warm-up incl. graph capture
    full = min(run(mode, a.new) for _ in range(a.iters))
    pre = min(run(mode, 1) for _ in range(a.iters))
    dec = full - pre
    print(json.dumps({"mode": mode, "model": a.model, "layers": cfg.num_hidden_layers, "batch": a.batch,
                      "prefix": a.prefix, "new_tokens": a.new, "total_s": full, "prefill_s": pre,
                      "decode_s": dec, "decode_tokens_per_s": a.batch * (a.new - 1) / dec,
                      "ms_per_decode_step": dec / (a.new - 1) * 1e3, "graph": not a.no_graph}))
