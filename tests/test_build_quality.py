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
    # 1 wave per unit: 4 waves / SIMD; the 4-waves-per-unit variant only runs when the grid cannot fill the chip anyway
    "suffix_attn_gqa.hip": [(r"suffix_attn_gqa_kernel", 256), (r"suffix_attn_gqa_kernelINS_\w+ELi\d+ELi1EE", 128)],
    "suffix_attn.hip": [(r"suffix_attn_kernel", 512), (r"suffix_attn_kernelINS_\w+ELi\d+ELi1ELi1E", 80)],  # MHA decode: 6 waves / SIMD
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
    assert "v_mfma_f32_32x32x16_bf16 a[0:15]" in out and "a[128:131]" in out
