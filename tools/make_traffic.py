#!/usr/bin/env python3
"""Turn a profiles/<tag>_pmc.txt table (tools/profile_round.sh) into profiles/traffic_latest.json, the per-launch
HBM traffic bench.py reports as roofline.traffic.  FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B,
MI355X_MICROARCH.md HBM section); both counters are in KB."""
import json
import sys


def main(path, tag):
    vals = {}
    for line in open(path):
        parts = line.split()
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c in parts:
                i = parts.index(c)
                kern = "suffix" if "suffix_attn_kernel" in line else "prefix" if "prefix_attn" in line else None
                if kern:
                    vals[(kern, c)] = float(parts[i + 2])
    out = {
        "source": f"profiles/{tag}_pmc.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python bench.py "
        "--steps 128 --warmup 0 --no-cpu-baseline --no-nosharing`; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md",
    }
    for k in ("suffix", "prefix"):
        f, w = vals.get((k, "FETCH_SIZE")), vals.get((k, "WRITE_SIZE"))
        if f is not None and w is not None:
            out[f"{k}_hbm_bytes_per_launch"] = (2 * f + w) * 1024
            out[f"{k}_fetch_kb_raw"] = f
            out[f"{k}_write_kb"] = w
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
