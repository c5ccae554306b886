#!/usr/bin/env python3
"""Turn a profiles/<tag>_pmc.txt table (tools/profile_round.sh) into profiles/traffic_latest.json, the per-launch
HBM traffic bench.py reports as roofline.traffic -- only for the command line the counters were collected with
(steps, warm-up and shape are recorded here and compared by bench.py).  FETCH_SIZE is doubled (gfx950 counts 128-B
requests as 64 B, MI355X_MICROARCH.md HBM section); both counters are in KB."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main(path, tag, steps, warmup):
    from bench import launch_schedule

    B, P, S, Hq, Hkv, D, e = 1024, 2048, 128, 32, 32, 128, 2
    vals = {}
    for line in open(path):
        if "[set-up" in line:  # placement probes (tools/rocprof_summary.py): not launches of the schedule
            continue
        parts = line.split()
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c in parts:
                i = parts.index(c)
                kern = "suffix" if "suffix_attn" in line else "prefix" if "prefix_attn" in line else None
                if kern:
                    vals[(kern, c)] = float(parts[i + 2])
    launches = launch_schedule(steps, warmup, 0, S)  # the profiled command runs with --trials 0
    suf_alg = sum(2 * e * Hkv * D * B * s + 2 * B * Hq * D * e + 4 * B * Hq for s in launches) / len(launches)
    pre_alg = 2 * P * Hkv * D * e + 2 * B * Hq * D * e + 4 * B * Hq
    out = {
        "source": f"profiles/{tag}_pmc.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python bench.py "
        f"--steps {steps} --warmup {warmup} --trials 0` (untimed legs off); per-launch mean over all {len(launches)} launches of the run "
        "(capture warm-ups, warm-up steps, timed steps: bench.launch_schedule); FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md",
        "steps": steps, "warmup": warmup, "batch": B, "prefix": P, "max_suffix": S, "qheads": Hq, "kvheads": Hkv,
        "suffix_algorithmic_bytes_per_launch": suf_alg, "prefix_algorithmic_bytes_per_launch": pre_alg,
    }
    for k in ("suffix", "prefix"):
        f, w = vals.get((k, "FETCH_SIZE")), vals.get((k, "WRITE_SIZE"))
        if f is not None and w is not None:
            out[f"{k}_hbm_bytes_per_launch"] = (2 * f + w) * 1024
            out[f"{k}_fetch_kb_raw"] = f
            out[f"{k}_write_kb"] = w
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
