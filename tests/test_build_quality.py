"""-m "not gpu": register allocation of the hot kernels, checked at compile time (hipcc cross-compiles gfx950
here).  A spill inside the prefix loop costs 2x (scratch traffic shares vmcnt with the LDS-DMA pipeline), and
it appears or disappears with small source changes, so the build is pinned: no scratch in any kernel of the
product path, and the occupancy each kernel was designed for."""
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parent.parent / "hydragen_amd" / "csrc"
HIPCC = "/opt/rocm/bin/hipcc"
# file -> (max VGPRs per kernel matching the regex)
LIMITS = {
    "prefix_attn_w64.hip": [(r"prefix_attn_w64_kernel", 512)],   # 1 wave / SIMD: the unified count (VGPRs + 192 AGPRs)
    # 2 waves / SIMD (two 8-KiB V tiles per wave bound the occupancy anyway); the count includes the 64 AGPRs of the K sets
    "suffix_attn_gqa.hip": [(r"suffix_attn_gqa_kernel", 256)],
    "suffix_attn.hip": [(r"suffix_attn_kernel", 512), (r"suffix_attn_kernelINS_\w+ELi\d+ELi1ELi1ELi\dE", 80)],  # <T, D, R = 1, WPU = 1, NPRE>, MHA decode: 6 waves / SIMD
    "combine.hip": [(r"combine", 128)],
    "rope_append.hip": [(r"rope_append", 128)],
}


_ASM_CACHE = {}


def _device_asm(src: str) -> str:
    """gfx950 assembly of one source file (compiled once per test session: the prefix kernel takes ~90 s)."""
    if src not in _ASM_CACHE:
        _ASM_CACHE[src] = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                                          str(CSRC / src), "-o", "-"], capture_output=True, text=True, check=True).stdout
    return _ASM_CACHE[src]


def _metadata(src: str):
    out = _device_asm(src)
    kernels = []
    for blk in out.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        kernels.append(dict(name=name,
                            vgpr=int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)),
                            spill=int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)),
                            sspill=int(re.search(r"\.sgpr_spill_count:\s+(\d+)", blk).group(1)),
                            scratch=int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))))
    return src, kernels


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_hot_kernels_have_no_scratch_and_keep_their_occupancy():
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(_metadata, LIMITS))
    for src, kernels in results:
        assert kernels, src
        for k in kernels:
            assert k["spill"] == 0 and k["scratch"] == 0, f"{src}: {k}"
            for pat, lim in LIMITS[src]:
                if re.search(pat, k["name"]):
                    assert k["vgpr"] <= lim, f"{src}: {k['name']} uses {k['vgpr']} VGPRs (> {lim})"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_prefix_kernel_owns_its_accumulator_registers():
    """prefix_attn_w64.hip keeps the O accumulators and the Q fragments in literal AGPRs (a[0:191]) that only its own
    inline-asm statements name.  That is safe only while hipcc itself never touches an AGPR in those kernels (it would,
    for spills): no instruction outside ;;#ASMSTART / ;;#ASMEND may name one, and every kernel must allocate >= 160."""
    out = _device_asm("prefix_attn_w64.hip")
    inasm, bad = False, []
    for line in out.splitlines():
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
        elif t.startswith(";;#ASMEND"):
            inasm = False
        elif not inasm and t and not t.startswith((";", ".", "//")) and re.search(r"\ba\[?\d", t.split(";")[0]):
            bad.append(t)
    assert not bad, bad[:5]
    counts = [int(x) for x in re.findall(r"\.agpr_count:\s+(\d+)", out)]
    assert counts and min(counts) >= 160, counts
    # (register numbers are assembler expressions of template constants; hipcc prints the larger ones in hex)
    assert "v_mfma_f32_32x32x16_bf16 a[0:15]" in out and re.search(r"a\[(128|0x80):(131|0x83)\]", out)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_gqa_suffix_kernel_owns_its_k_registers_while_loads_are_in_flight():
    """suffix_attn_gqa.hip loads the K fragments of step i+1 into literal AGPRs (a[0:63]) while step i is computed.
    Between the first and the last asm statement that names an AGPR (the pipelined loop) no compiler-generated
    instruction may name one: hipcc sees those registers as free between two clobbering statements and could park a
    value where a load is about to land.  (Outside the loop -- the 4-wave merge -- it may use them.)"""
    out = _device_asm("suffix_attn_gqa.hip")
    kernels, cur, inasm = {}, None, False
    for n, line in enumerate(out.splitlines()):
        t = line.strip()
        m = re.match(r"^(_ZN3hyd\w+):", t)
        if m:
            cur = m.group(1)
            kernels[cur] = dict(asm=[], comp=[])
        if cur is None:
            continue
        if t.startswith(";;#ASMSTART"):
            inasm = True
        elif t.startswith(";;#ASMEND"):
            inasm = False
        elif t and not t.startswith((";", ".", "//")) and re.search(r"\ba\[?\d", t.split(";")[0]):
            kernels[cur]["asm" if inasm else "comp"].append((n, t))
    assert len(kernels) == 8, sorted(kernels)
    for name, v in kernels.items():
        assert v["asm"], name
        lo, hi = v["asm"][0][0], v["asm"][-1][0]
        inside = [t for n, t in v["comp"] if lo <= n <= hi]
        assert not inside, f"{name}: compiler-generated AGPR use inside the pipelined loop: {inside[:4]}"
