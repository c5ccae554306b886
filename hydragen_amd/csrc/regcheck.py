"""Register-ownership checks on the compiled prefix kernels, run by build.py on every (re)compile and by tests/test_build_quality.py.

The prefix pass keeps its O accumulators and Q fragments in LITERAL registers that only its own inline-asm statements name
(prefix_unit_w64.h: RegsV = v[160:255] of the 8-wave kernels, RegsA = a[0:191] of the 4-wave kernels).  Nothing in the
language tells hipcc those registers are taken; what keeps it out is how this hipcc allocates (amdgpu_num_vgpr(80) on a
kernel without accumulator registers = at most 160 architectural VGPRs; no spills, hence no use for AGPRs).  A different
compiler release may allocate differently and would then corrupt O and Q silently, so the invariant is enforced where the
library is built: `check_prefix_asm` raises, build.py then produces no library, and loading fails loudly (no fallback).

Input: the device assembly hipcc prints with -S (it keeps the ;;#ASMSTART / ;;#ASMEND markers around inline asm; a
disassembled code object does not)."""
from __future__ import annotations

import re


class RegisterOwnershipError(RuntimeError):
    pass


def _kernels_meta(asm: str):
    out = []
    for blk in asm.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out.append(dict(name=name, agpr=int(blk.split()[0]),
                        vgpr=int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)),
                        spill=int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)),
                        scratch=int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))))
    return out


def eight_wave_usage(asm: str):
    """{kernel: (highest VGPR a compiler-generated instruction names, number of compiler-generated AGPR uses)} for the
    w64x8 kernels of one translation unit."""
    kern, inasm, hi, agpr = None, False, {}, {}
    for line in asm.splitlines():
        t = line.strip()
        m = re.match(r"^(_ZN3hyd\w+):", t)
        if m:
            kern = m.group(1) if "w64x8" in m.group(1) else None
            if kern:
                hi[kern], agpr[kern] = -1, 0
            continue
        if kern is None:
            continue
        if t.startswith(";;#ASMSTART"):
            inasm = True
        elif t.startswith(";;#ASMEND"):
            inasm = False
        elif not inasm and t and not t.startswith((";", ".", "//")):
            code = t.split(";")[0]
            for mm in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", code):
                hi[kern] = max(hi[kern], int(mm.group(2)) if mm.group(2) else int(mm.group(3)))
            agpr[kern] += bool(re.search(r"\ba\[?\d", code))
    return {k: (hi[k], agpr[k]) for k in hi}


def four_wave_compiler_agpr_uses(asm: str):
    """Compiler-generated instructions (outside asm statements) that name an accumulator register, anywhere in the unit."""
    inasm, bad = False, []
    for line in asm.splitlines():
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
        elif t.startswith(";;#ASMEND"):
            inasm = False
        elif not inasm and t and not t.startswith((";", ".", "//")) and re.search(r"\ba\[?\d", t.split(";")[0]):
            bad.append(t)
    return bad


def check_prefix_asm(asm: str, what: str = "prefix kernels") -> None:
    """Raise RegisterOwnershipError unless every prefix kernel of this translation unit keeps the contract."""
    meta = _kernels_meta(asm)
    if not meta:
        raise RegisterOwnershipError(f"{what}: no kernel metadata found in the assembly (did hipcc's -S output change?)")
    for k in meta:
        if k["spill"] or k["scratch"]:
            raise RegisterOwnershipError(f"{what}: {k['name']} spills ({k['spill']} VGPRs, {k['scratch']} bytes of scratch)")
        if "w64x8" in k["name"] and (k["agpr"] != 0 or k["vgpr"] != 256):
            raise RegisterOwnershipError(f"{what}: {k['name']} must allocate 256 VGPRs and no AGPR, has {k['vgpr']} / {k['agpr']}")
        if "prefix_attn_w64_kernel" in k["name"] and k["agpr"] < 192:
            raise RegisterOwnershipError(f"{what}: {k['name']} allocates {k['agpr']} AGPRs (< 192: the unit's asm names a[0:191])")
    use = eight_wave_usage(asm)
    if not use:
        raise RegisterOwnershipError(f"{what}: no 8-wave kernel found")
    for k, (hi, ag) in use.items():
        if not (0 <= hi < 160) or ag:
            raise RegisterOwnershipError(f"{what}: hipcc uses v{hi} / {ag} AGPR instructions in {k}: v[160:255] belong to the asm statements")
    bad = four_wave_compiler_agpr_uses(asm)
    if bad:
        raise RegisterOwnershipError(f"{what}: compiler-generated AGPR use next to asm-owned a[0:191]: {bad[:3]}")
