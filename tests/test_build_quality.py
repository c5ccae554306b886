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
    # 1 wave / SIMD: the unified count (VGPRs + 192 AGPRs); the 8-wave kernels (2 waves / SIMD) claim all 256 VGPRs of a wave
    "prefix_attn_w64.hip": [(r"prefix_attn_w64_kernel", 512), (r"prefix_attn_w64x8_kernel", 256)],
    "prefix_attn_w64_f16.hip": [(r"prefix_attn_w64_kernel", 512), (r"prefix_attn_w64x8_kernel", 256)],  # the fp16 instantiations
    # 2 waves / SIMD: K and V fragments of a step live in registers (one K and one V landing tile of LDS per wave)
    "suffix_attn_gqa.hip": [(r"suffix_attn_gqa_kernel", 256)],
    # <T, D, R = 1, WPU = 1, NPRE, U, OCC>, one-unit-per-wave MHA decode: 6 waves / SIMD; the token-row kernel: 4 waves / SIMD (8 tokens x 2 tensors, requests rotated)
    "suffix_attn.hip": [(r"suffix_attn_kernel", 512), (r"suffix_attn_kernelINS_\d\w+?ELi(64|128|256)ELi1ELi1ELi\dELi\dELi\d(ELi\d)?EEv", 80),
                        (r"suffix_attn_rows_kernel", 128)],
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
            d256_gqa = src == "suffix_attn_gqa.hip" and "ELi256E" in k["name"]  # one wave per SIMD; hipcc parks 3 values in AGPRs, nothing in memory
            assert (k["spill"] == 0 or d256_gqa) and k["scratch"] == 0, f"{src}: {k}"
            for pat, lim in LIMITS[src]:
                if re.search(pat, k["name"]):
                    assert k["vgpr"] <= (512 if d256_gqa else lim), f"{src}: {k['name']} uses {k['vgpr']} VGPRs (> {lim})"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_prefix_kernel_owns_its_accumulator_registers():
    """prefix_attn_w64.hip keeps the O accumulators and the Q fragments in literal AGPRs (a[0:191]) that only its own
    inline-asm statements name.  That is safe only while hipcc itself never touches an AGPR in those kernels (it would,
    for spills): no instruction outside ;;#ASMSTART / ;;#ASMEND may name one, and every kernel must allocate >= 160."""
    out = _device_asm("prefix_attn_w64.hip") + _device_asm("prefix_attn_w64_f16.hip")  # bf16 / fp16 instantiations
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
    counts = [int(blk.split()[0]) for blk in out.split("  - .agpr_count:")[1:]
              if "prefix_attn_w64_kernel" in re.search(r"\.name:\s+(\S+)", blk).group(1)]
    assert counts and min(counts) >= 160, counts
    # (register numbers are assembler expressions of template constants; hipcc prints the larger ones in hex)
    assert "v_mfma_f32_32x32x16_bf16 a[0:15]" in out and "v_mfma_f32_32x32x16_f16 a[0:15]" in out and re.search(r"a\[(128|0x80):(131|0x83)\]", out)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_eight_wave_prefix_kernel_owns_the_top_of_the_register_file():
    """The 8-wave prefix kernels (two waves per SIMD) keep O in the literal VGPRs v[160:223] and Q in v[224:255], named
    only inside their asm statements (prefix_unit_w64.h, RegsV).  hipcc is kept below v160 by amdgpu_num_vgpr(80) on a
    kernel that uses no accumulator register: no compiler-generated instruction may name v160 or above, the kernels
    allocate no AGPR (hipcc would otherwise be held to 128 VGPRs and park values in accumulator registers), and the
    descriptor covers the whole 256-register file."""
    out = _device_asm("prefix_attn_w64.hip") + _device_asm("prefix_attn_w64_f16.hip")
    kern, inasm, hi, agpr = None, False, {}, {}
    for line in out.splitlines():
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
    assert len(hi) == 8, sorted(hi)  # {bf16, f16} x {dense, causal} x {128-row, 256-row workgroups}
    for k in hi:
        assert 0 <= hi[k] < 160 and agpr[k] == 0, (k, hi[k], agpr[k])
    for blk in out.split("  - .agpr_count:")[1:]:
        if "w64x8" in re.search(r"\.name:\s+(\S+)", blk).group(1):
            assert int(blk.split()[0]) == 0 and int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) == 256, blk[:200]
    assert re.search(r"v_mfma_f32_32x32x16_bf16 v\[(160|0xa0):(175|0xaf)\]", out) and re.search(r"v\[(224|0xe0):(227|0xe3)\]", out)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_build_refuses_a_compiler_that_touches_the_literal_registers():
    """csrc/regcheck.py is what build.py runs on the assembly of every (re)compile of the prefix kernels: the real
    assembly passes; the same text with ONE compiler-looking instruction on v200 planted in an 8-wave kernel, with one on an AGPR in a
    4-wave kernel, or with a spill in the metadata, is refused -- build.py then deletes objects, stamps and library."""
    import sys
    sys.path.insert(0, str(CSRC))
    from regcheck import RegisterOwnershipError, check_prefix_asm

    asm = _device_asm("prefix_attn_w64.hip")
    check_prefix_asm(asm, "as compiled")
    label = re.search(r"^(_ZN3hyd\w*w64x8\w+):", asm, flags=re.M)
    planted = asm[:label.end()] + "\n\tv_mov_b32_e32 v200, v1\n" + asm[label.end():]
    with pytest.raises(RegisterOwnershipError, match="v200"):
        check_prefix_asm(planted, "planted VGPR")
    label4 = re.search(r"^(_ZN3hyd22prefix_attn_w64_kernel\w+):", asm, flags=re.M)
    planted = asm[:label4.end()] + "\n\tv_accvgpr_write_b32 a5, v1\n" + asm[label4.end():]
    with pytest.raises(RegisterOwnershipError, match="AGPR"):
        check_prefix_asm(planted, "planted AGPR")
    with pytest.raises(RegisterOwnershipError, match="spills"):
        check_prefix_asm(asm.replace(".vgpr_spill_count: 0", ".vgpr_spill_count: 2", 1), "planted spill")


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_gqa_suffix_kernel_names_no_register_by_hand():
    """suffix_attn_gqa.hip streams K and V through LDS landing tiles (LDS-DMA: no destination register) and issues every load that
    lands in registers as plain C++: no accumulator register is allocated at all (round 4's form kept K sets in asm-owned AGPRs and
    needed an audit of who touches them), nothing spills, and two waves per SIMD fit."""
    out = _device_asm("suffix_attn_gqa.hip")
    metas = []
    for blk in out.split("  - .agpr_count:")[1:]:
        metas.append((re.search(r"\.name:\s+(\S+)", blk).group(1), int(blk.split()[0]), int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)),
                      int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)), int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))))
    # {f16, bf16} x ({64, 128} x ({1, 4 waves per unit} x {non-temporal K/V or not} + {2, 4 kv heads per workgroup}) + 256 x {1, 2 kv heads per workgroup})
    assert len(metas) == 28, [m[0] for m in metas]
    for name, agpr, vgpr, spill, scratch in metas:
        if "ELi256E" in name:  # one wave per SIMD: hipcc parks a few values in accumulator registers of ITS choosing; nothing in memory
            assert scratch == 0 and vgpr <= 512, (name, agpr, vgpr, spill, scratch)
        else:
            assert agpr == 0 and spill == 0 and scratch == 0 and vgpr <= 256, (name, agpr, vgpr, spill, scratch)
    # the loop's only vector-memory waits are the hand-placed full drains: hipcc must not add counted waits of its own between the
    # DMA issue and the arithmetic (it cannot see the DMAs; a counted wait there would serialise the stream)
    for m in re.finditer(r"^(_ZN3hyd22suffix_attn_gqa_kernel\w+):(.*?)s_endpgm", out, flags=re.S | re.M):
        body = m.group(2)
        first_dma = body.find(" lds")
        assert first_dma > 0, m.group(1)
        # ... and none between the request of q / the first partials and the first DMA (they ride under the stream)
        # (checked for head dim 128, the configuration every BASELINE shape uses: at 256 hipcc moves the requested values into its
        # accumulator registers, which takes the wait; at 64 its register reuse puts one counted wait in the 16-bit partial's branch)
        q_load = body.find("global_load_dwordx4")
        assert 0 < q_load < first_dma, m.group(1)
        if "ELi128E" in m.group(1):
            # (a counted wait that leaves the 7 youngest requests in flight is tolerated: since round 6 hipcc's register reuse puts one
            # into the branch of a SECOND 16-bit partial -- a two-level hierarchy on grouped-query heads --, where it waits for the
            # query row and the first LSE, the oldest requests of the wave; the one-partial and split-slice paths have none)
            early = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body[q_load:first_dma])]
            assert all(n >= 7 for n in early) and len(early) <= 1, (m.group(1), early, "hipcc waits for q or a partial in front of the K/V stream")
        lines = body[first_dma:].splitlines()
        last_dma = max(i for i, ln in enumerate(lines) if ln.rstrip().endswith(" lds") or " lds " in ln)
        counted = [ln.strip() for ln in lines[:last_dma] if re.search(r"s_waitcnt vmcnt\((?!0\))", ln)]
        if "ELi64E" in m.group(1):  # head dim 64 keeps two key steps in flight: its hand-placed wait leaves the younger step's 2 x 4 requests out
            counted = [ln for ln in counted if ln != "s_waitcnt vmcnt(8)"]
        assert not counted, (m.group(1), counted[:3])


def _sregs(tok: str):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def valu_sgpr_to_vmem_hazards(asm: str, need: int = 5):
    """(kernel, buffer instruction, writer) for every buffer_load / buffer_store whose scalar operands (resource, offset)
    were written by a VALU instruction (v_readlane / v_readfirstlane) fewer than `need` wait states earlier -- the gfx9
    hazard hipcc pads for its own instructions but cannot see when the memory instruction sits inside an asm statement."""
    kern, window, bad = None, [], []
    for ln in asm.splitlines():
        s = ln.strip()
        if s.startswith("_Z") and ":" in s:
            kern, window = s.split(":")[0], []
            continue
        if not s or s[0] in ";./":
            continue
        op = s.split()[0]
        if op.startswith(("buffer_load", "buffer_store")):
            used = set()
            for tok in s.split(None, 1)[1].replace(",", " ").split():
                used |= _sregs(tok)
            ws = 0
            for pop, pdst in reversed(window):
                if ws >= need:
                    break
                if pdst & used:
                    bad.append((kern, s, pop))
                    break
                m = re.fullmatch(r"s_nop (\d+)", pop)
                ws += int(m.group(1)) + 1 if m else 1
        dst = _sregs(s.split(None, 1)[1].split(",")[0]) if op in ("v_readlane_b32", "v_readfirstlane_b32") else set()
        window = (window + [(s if op == "s_nop" else op, dst)])[-12:]
    return bad


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["prefix_attn_w64.hip", "prefix_attn_w64_f16.hip", "suffix_attn_gqa.hip", "suffix_attn.hip"])
def test_asm_memory_instructions_keep_their_distance_from_valu_written_scalars(src):
    """Found the hard way: the persistent prefix kernel restores spilled scalars with v_readlane right in front of the
    LDS-DMA asm statements; without wait states the DMA read a stale offset (timing-dependent garbage in the ragged
    16384-key fixture).  Checked on the compiled code of every instantiation."""
    assert valu_sgpr_to_vmem_hazards(
        "_Zk:\nv_readlane_b32 s5, v1, 3\ns_nop 1\nbuffer_load_dwordx4 v1, s[8:11], s5 offen lds\n"), "the checker sees a planted hazard"
    assert not valu_sgpr_to_vmem_hazards(
        "_Zk:\nv_readlane_b32 s5, v1, 3\ns_nop 4\nbuffer_load_dwordx4 v1, s[8:11], s5 offen lds\n")
    bad = valu_sgpr_to_vmem_hazards(_device_asm(src))
    assert not bad, f"{len(bad)} hazards, first: {bad[:3]}"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_prefix_kernels_leave_m0_to_the_dma_statements():
    """The prefix pass writes M0 once per tensor and block and addresses the block's pieces through immediate offsets
    (prefix_unit_w64.h, dma_m0 / dma16w): between that write and the block's last LDS-DMA no compiler-generated instruction
    may touch M0.  Checked the strong way: outside the asm statements nothing in these kernels names m0 at all."""
    inasm, kern, bad, writes = False, None, [], 0
    for ln in (_device_asm("prefix_attn_w64.hip") + _device_asm("prefix_attn_w64_f16.hip")).splitlines():
        t = ln.strip()
        if t.startswith("_Z") and ":" in t:
            kern = t.split(":")[0]
        elif t.startswith(";;#ASMSTART"):
            inasm = True
        elif t.startswith(";;#ASMEND"):
            inasm = False
        elif re.search(r"\bm0\b", t.split(";")[0]):
            if inasm:
                writes += t.startswith("s_mov_b32 m0")
            else:
                bad.append((kern, t))
    assert writes > 100, "the DMA statements' M0 writes were not found: did the asm markers change?"
    assert not bad, bad[:3]
