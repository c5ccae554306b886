"""Development tool: one-line summary of a bench.py JSON line.  `python tools/show_bench.py FILE` (or `-` for stdin)."""
import json, sys
if len(sys.argv) < 2:
    sys.exit(__doc__)
text = sys.stdin.read() if sys.argv[1] == "-" else open(sys.argv[1]).read()
d = json.loads(text.strip().splitlines()[-1])
other = d.get("roofline_other") or d.get("roofline_prefix") or {}
roofs = {r.get("bound"): r for r in (d["roofline"], other) if r}  # (the dominant kernel is the suffix pass at C2, the prefix pass at C5)
print(sys.argv[1], {k: round(d[k], 1) for k in ("value", "attn_us_per_step", "prefix_us", "suffix_us_mean") if k in d},
      "hbm_frac", round(roofs.get("hbm", {}).get("frac", 0.0), 3), "mfma_frac", round(roofs.get("mfma", {}).get("frac", 0.0), 3))
