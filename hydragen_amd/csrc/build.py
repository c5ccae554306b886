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


def _includes(src: str, seen=None) -> list:
    """The project headers a source includes, transitively (so that a kernel's object is rebuilt for ITS headers only)."""
    import re

    seen = set() if seen is None else seen
    for inc in re.findall(r'^\s*#include\s+"([^"]+)"', (HERE / src).read_text(), flags=re.M):
        path = (HERE / src).parent / inc
        rel = os.path.relpath(path.resolve(), HERE)
        if path.exists() and rel not in seen:
            seen.add(rel)
            _includes(rel, seen)
    return sorted(seen)


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any((HERE / d).stat().st_mtime > t for d in deps)


def _compile(src: str, force: bool) -> Path:
    obj = HERE / (Path(src).stem + ".o")
    if force or _stale(obj, [src] + _includes(src) + ["build.py"]):
        cmd = [HIPCC, *FLAGS, "-c", str(HERE / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


# The prefix kernels keep O and Q in literal registers that only their asm statements name (prefix_unit_w64.h); whether
# hipcc stays out of them is a property of the compiler release, so it is checked on the assembly of every (re)compile
# (regcheck.py).  A failed check leaves no object, no stamp and no library behind: nothing loads, nothing computes wrongly.
REGCHECKED = ["prefix_attn_w64.hip", "prefix_attn_w64_f16.hip"]


def _stamp(src: str) -> Path:
    return HERE / (Path(src).stem + ".regcheck")


def _regcheck(src: str, force: bool) -> None:
    from regcheck import check_prefix_asm  # (build.py runs as a script and as hydragen_amd.csrc.build: plain import by path)

    stamp = _stamp(src)
    if not (force or _stale(stamp, [src] + _includes(src) + ["build.py", "regcheck.py"])):
        return
    stamp.unlink(missing_ok=True)
    cmd = [HIPCC, *[f for f in FLAGS if f not in ("-fPIC", "-Werror")], "-S", "--cuda-device-only", str(HERE / src), "-o", "-"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr}")
    check_prefix_asm(r.stdout, src)
    ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()
    stamp.write_text("registers checked: " + (ver[0] if ver else "hipcc") + "\n")


def build(force: bool = False, verbose: bool = True) -> Path:
    if str(HERE) not in sys.path:
        sys.path.insert(0, str(HERE))
    with ThreadPoolExecutor(max_workers=6) as ex:
        checks = [ex.submit(_regcheck, s, force) for s in REGCHECKED]
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
        try:
            for c in checks:
                c.result()
        except Exception:
            for s in REGCHECKED:  # nothing built from these sources may survive
                (HERE / (Path(s).stem + ".o")).unlink(missing_ok=True)
                _stamp(s).unlink(missing_ok=True)
            LIB.unlink(missing_ok=True)
            raise
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
