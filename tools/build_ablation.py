#!/usr/bin/env python3
"""Development tool: build an ablation library (HYD_ABLATION_BUILD: A/B environment switches, timing-ablation kernel
variants) into build_probe/libhydragen_abl.so without touching the product
objects.  Use it with HYDRAGEN_HIP_LIB=build_probe/libhydragen_abl.so."""
import subprocess, sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
src, out = REPO / "hydragen_amd" / "csrc", REPO / "build_probe"
out.mkdir(exist_ok=True)
srcs = ["api.hip", "prefix_attn_w64.hip", "prefix_attn_w64_f16.hip", "suffix_attn.hip", "suffix_attn_gqa.hip", "combine.hip", "rope_append.hip", "layer_ops.hip", "allreduce.hip"]
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DHYD_ABLATION_BUILD", "-Wno-unused-function"]


sys.path.insert(0, str(src))
from build import _includes  # noqa: E402  (the product build's include scanner: rebuild an object for ITS headers only)


def cc(f):
    o = out / (Path(f).stem + ".o")
    if o.exists() and o.stat().st_mtime > max([(src / f).stat().st_mtime] + [(src / h).stat().st_mtime for h in _includes(f)]):
        return str(o)
    r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", str(src / f), "-o", str(o)], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    return str(o)


with ThreadPoolExecutor(4) as ex:
    objs = list(ex.map(cc, srcs))
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(out / "libhydragen_abl.so")], capture_output=True, text=True)
sys.exit(r.stderr[-2000:] if r.returncode else 0)
