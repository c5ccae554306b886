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
    "prefix_attn_pl.hip": [(r"prefix_attn_pl_kernel", 256)],   # 2 waves / SIMD
    "suffix_attn_gqa.hip": [(r"suffix_attn_gqa_kernel", 128)],  # 4 waves / SIMD
    "suffix_attn.hip": [(r"suffix_attn_kernel", 512), (r"suffix_attn_kernelINS_\w+ELi\d+ELi1ELi1E", 80)],  # MHA decode: 6 waves / SIMD
    "combine.hip": [(r"combine", 128)],
    "rope_append.hip": [(r"rope_append", 128)],
}


def _metadata(src: str):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                          str(CSRC / src), "-o", "-"], capture_output=True, text=True, check=True).stdout
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
