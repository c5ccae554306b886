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


# The prefix kernels keep O and Q in literal registers that only their asm statements name (prefix_unit_w64.h); whether
# hipcc stays out of them is a property of the compiler release, so it is checked on the assembly of every (re)compile
# (regcheck.py) -- the assembly OF THE OBJECT THAT IS LINKED: these sources are compiled with -save-temps=obj into a scratch
# directory, the device assembly of that very compile (same flags, -fPIC and -Werror included) is checked, and only then is
# the object moved into place.  A failed check leaves no object, no stamp and no library behind: nothing loads, nothing
# computes wrongly.
REGCHECKED = ["prefix_attn_w64.hip", "prefix_attn_w64_f16.hip"]


def _stamp(src: str) -> Path:
    return HERE / (Path(src).stem + ".regcheck")


def _compile(src: str, force: bool) -> Path:
    obj = HERE / (Path(src).stem + ".o")
    checked = src in REGCHECKED
    deps = [src] + _includes(src) + ["build.py"]
    stale = force or _stale(obj, deps) or (checked and _stale(_stamp(src), deps + ["regcheck.py"]))
    if not stale:
        return obj
    if not checked:
        cmd = [HIPCC, *FLAGS, "-c", str(HERE / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj
    import shutil
    import tempfile

    from regcheck import check_prefix_asm  # (build.py runs as a script and as hydragen_amd.csrc.build: plain import by path)

    _stamp(src).unlink(missing_ok=True)
    obj.unlink(missing_ok=True)
    tmp = Path(tempfile.mkdtemp(prefix="hyd_regcheck_"))
    try:
        tobj = tmp / obj.name
        r = subprocess.run([HIPCC, *FLAGS, "-save-temps=obj", "-c", str(HERE / src), "-o", str(tobj)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        asm = sorted(tmp.glob(Path(src).stem + "-hip-amdgcn-amd-amdhsa-gfx950*.s"))
        if len(asm) != 1:
            raise RuntimeError(f"{src}: expected one device assembly file from -save-temps, found {[a.name for a in asm]}")
        check_prefix_asm(asm[0].read_text(), src)
        shutil.move(str(tobj), str(obj))
        ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()
        _stamp(src).write_text("registers checked on the linked object's own assembly: " + (ver[0] if ver else "hipcc") + "\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    if str(HERE) not in sys.path:
        sys.path.insert(0, str(HERE))
    with ThreadPoolExecutor(max_workers=6) as ex:
        try:
            objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
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
