import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", {k: round(d[k], 1) for k in ("value", "attn_us_per_step", "prefix_us", "suffix_us_mean")},
      "hbm_frac", round(d["roofline"]["frac"], 3), "mfma_frac", round(d["roofline_prefix"]["frac"], 3))
