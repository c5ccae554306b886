"""Build libhydragen_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python hydragen_amd/csrc/build.py [--force]

Object files are compiled in parallel, one per .hip source, then linked into
hydragen_amd/csrc/libhydragen_hip.so (git-ignored; it travels to the GPU box with the tree).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
SOURCES = ["api.hip", "prefix_attn_w64.hip", "prefix_attn_w64_f16.hip", "suffix_attn.hip", "suffix_attn_gqa.hip", "combine.hip", "rope_append.hip", "layer_ops.hip", "allreduce.hip"]
HEADERS = sorted(h.name for h in HERE.glob("*.h")) + ["../../include/hydragen_hip.h"]
LIB = HERE / "libhydragen_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Werror", "-Wno-unused-function"]
if os.environ.get("HYD_ABLATION_BUILD"):  # development only: A/B switches, timing-ablation kernel variants
    FLAGS.append("-DHYD_ABLATION_BUILD")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any((HERE / d).stat().st_mtime > t for d in deps)


def _compile(src: str, force: bool) -> Path:
    obj = HERE / (Path(src).stem + ".o")
    if force or _stale(obj, [src] + HEADERS + ["build.py"]):
        cmd = [HIPCC, *FLAGS, "-c", str(HERE / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
