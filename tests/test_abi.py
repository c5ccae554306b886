"""-m "not gpu": the C-ABI library loads, exports every symbol include/hydragen_hip.h declares, and its
host-side logic (validation, planning, workspace sizing, error strings) behaves -- no compute calls."""
import ctypes as C
import re
from pathlib import Path

import pytest

from hydragen_amd import _lib
from hydragen_amd._lib import DecodeParams, PrefixParams, SuffixParams

REPO = Path(__file__).resolve().parent.parent


def test_exports_match_header():
    lib = _lib.load()
    header = (REPO / "include" / "hydragen_hip.h").read_text()
    declared = set(re.findall(r"\b(hyd_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    assert lib.hyd_version() == 500 == _lib.ABI_VERSION


def test_dynamic_symbol_table_is_exactly_the_header():
    """The product library is built with hidden visibility: `nm -D` must list the header's entry points and nothing
    else of ours (no measurement exports, no C++ launchers), and the library must not read the environment."""
    import subprocess
    header = (REPO / "include" / "hydragen_hip.h").read_text()
    declared = set(re.findall(r"\b(hyd_[a-z_0-9]+)\s*\(", header))
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(_lib.lib_path())], text=True)
    defined = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert defined == declared, defined ^ declared
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", str(_lib.lib_path())], text=True)
    assert "getenv" not in undefined, "the product library must not read environment variables"


def test_struct_layout_matches_c():
    """ctypes mirrors must have the C compiler's sizes (checked against a gcc build of the header)."""
    import subprocess, tempfile
    src = '#include "hydragen_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(hyd_prefix_params), sizeof(hyd_partial), sizeof(hyd_suffix_params), sizeof(hyd_level), sizeof(hyd_decode_params), sizeof(hyd_rope_params), sizeof(hyd_add_rmsnorm_params), sizeof(hyd_swiglu_params), sizeof(hyd_sample_params));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "s.c").write_text(src)
        subprocess.check_call(["gcc", "-I", str(REPO / "include"), str(Path(d) / "s.c"), "-o", str(Path(d) / "s")])
        sizes = list(map(int, subprocess.check_output([str(Path(d) / "s")]).split()))
    assert sizes == [C.sizeof(PrefixParams), C.sizeof(_lib.Partial), C.sizeof(SuffixParams), C.sizeof(_lib.Level),
                     C.sizeof(DecodeParams), C.sizeof(_lib.RopeParams), C.sizeof(_lib.AddRmsnormParams),
                     C.sizeof(_lib.SwigluParams), C.sizeof(_lib.SampleParams)]


def _prefix(**kw):
    p = PrefixParams()
    p.dtype, p.B, p.nq, p.Hq, p.Hkv, p.D, p.sb, p.kv_len = 1, 1024, 1, 32, 32, 128, 1, 2048
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _plan(p):
    lib = _lib.load()
    ns, grid, sl = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.hyd_prefix_plan(C.byref(p), C.byref(ns), C.byref(grid), C.byref(sl))
    return rc, ns.value, grid.value, sl.value


def test_plan_c2_fills_the_chip_without_split():
    rc, ns, grid, sl = _plan(_prefix())
    assert (rc, ns, grid) == (0, 1, 256)  # 32 heads x 8 row blocks of 128 queries = 256 CUs
    assert _lib.load().hyd_prefix_workspace_bytes(C.byref(_prefix())) == 0


def test_plan_c3_splits_kv():
    # B=64, 32q/8kv, P=16384: 8 heads x 2 row blocks = 16 units -> split-KV (SURVEY 5 'build implication')
    p = _prefix(B=64, Hq=32, Hkv=8, kv_len=16384)
    rc, ns, grid, sl = _plan(p)
    assert rc == 0 and ns > 1 and grid == 16 * ns and sl % 128 == 0 and ns * sl >= 16384
    assert _lib.load().hyd_prefix_workspace_bytes(C.byref(p)) >= ns * 64 * 32 * 128 * 4


@pytest.mark.parametrize("kw,code,frag", [
    (dict(D=96), -2, "head_dim"),
    (dict(dtype=2), -2, "dtype"),
    (dict(B=1000, sb=3), -1, "not divisible"),
    (dict(Hq=32, Hkv=5), -1, "not divisible"),
    (dict(B=0), -1, "non-positive"),
])
def test_bad_arguments_are_rejected_with_a_message(kw, code, frag):
    lib = _lib.load()
    rc, *_ = _plan(_prefix(**kw))
    assert rc == code
    assert frag in lib.hyd_last_error_string().decode()
    # the launch entry point refuses the same arguments before touching the device
    assert lib.hyd_prefix_attn_fwd(C.byref(_prefix(**kw)), None) == code
    with pytest.raises((ValueError, NotImplementedError)):
        _lib.check(code)


def test_null_and_misaligned_pointers_rejected():
    lib = _lib.load()
    p = _prefix()
    assert lib.hyd_prefix_attn_fwd(C.byref(p), None) == -1 and "null" in lib.hyd_last_error_string().decode()
    s = SuffixParams()
    s.dtype, s.B, s.nq, s.Hq, s.Hkv, s.D, s.kv_len = 1, 4, 1, 8, 8, 128, 16
    s.q = 0x1008
    assert lib.hyd_suffix_attn_fwd(C.byref(s), None) == -1 and "aligned" in lib.hyd_last_error_string().decode()


def test_oversized_sequence_span_rejected_before_launch():
    lib = _lib.load()
    s = SuffixParams()
    s.dtype, s.B, s.nq, s.Hq, s.Hkv, s.D = 1, 1, 1, 8, 8, 128
    s.kv_len, s.k_tok_stride, s.v_tok_stride = 1 << 21, 1024, 1024          # 2^21 tokens * 2 KiB = 4 GiB per sequence
    s.k_head_stride = s.v_head_stride = 128
    s.q = s.k = s.v = s.out = 0x10000
    assert lib.hyd_suffix_attn_fwd(C.byref(s), None) == -2
    assert "2 GiB" in lib.hyd_last_error_string().decode()


def test_decode_workspace_accounting():
    lib = _lib.load()
    sb = (C.c_int32 * 2)(1, 32)
    ln = (C.c_int32 * 2)(1024, 64)
    rows = 1024 * 32
    # C4: two levels, both fill the chip -> one partial + one fp32 LSE vector per level, + a 16-bit partial and an LSE
    # vector for the unique pass's own (two-stream form).  The shape helper is an upper bound for every form of the call,
    # f32_partials included: an unsplit level's partial is sized in fp32.
    # (the second level is a small one -- it runs on the grouped-query kernel and keeps a 16-bit partial in every form)
    want = (rows * 128 * 4 + rows * 4) + 2 * (rows * 128 * 2 + rows * 4)
    assert lib.hyd_workspace_bytes(1024, 1, 32, 32, 128, 2, sb, ln) == want
    d = DecodeParams()
    d.suffix.dtype, d.suffix.B, d.suffix.nq, d.suffix.Hq, d.suffix.Hkv, d.suffix.D, d.suffix.kv_len = 1, 1024, 1, 32, 32, 128, 16
    d.n_levels = 2
    for i, (b_, l_) in enumerate(((1, 1024), (32, 64))):
        d.levels[i].sb, d.levels[i].kv_len = b_, l_
    for f32 in (0, 1):
        d.f32_partials = f32
        assert lib.hyd_decode_workspace_bytes(C.byref(d)) <= want
    assert lib.hyd_decode_workspace_bytes(C.byref(d)) == want  # f32_partials = 1: exactly the bound
    d = DecodeParams()
    d.suffix.dtype, d.suffix.B, d.suffix.nq, d.suffix.Hq, d.suffix.Hkv, d.suffix.D = 1, 1024, 1, 32, 32, 128
    d.n_levels = 9
    assert lib.hyd_decode_attn_fused(C.byref(d), None) == -1


def _decode(B, Hq, Hkv, D, levels, kv_len=16):
    d = DecodeParams()
    d.suffix.dtype, d.suffix.B, d.suffix.nq, d.suffix.Hq, d.suffix.Hkv, d.suffix.D = 1, B, 1, Hq, Hkv, D
    d.suffix.kv_len = kv_len
    d.n_levels = len(levels)
    for i, (sb, n) in enumerate(levels):
        d.levels[i].sb, d.levels[i].kv_len = sb, n
    return d


def test_prefix_only_early_exit_workspace_covers_the_split_slices():
    """one shared level + empty unique KV runs the prefix pass with its own split plan, so the decode
    workspace query must answer with what that pass needs (it used to return the small-level size)."""
    lib = _lib.load()
    d = _decode(4, 8, 8, 128, [(1, 1024)], kv_len=0)
    p = _prefix(B=4, Hq=8, Hkv=8, kv_len=1024)
    rc, ns, _, _ = _plan(p)
    assert rc == 0 and ns > 1
    want = lib.hyd_prefix_workspace_bytes(C.byref(p))
    assert want >= ns * 4 * 8 * 128 * 4
    assert lib.hyd_decode_workspace_bytes(C.byref(d)) == want


def test_deep_hierarchy_partials_fit_the_merge_budget():
    """three long levels with one kv head used to plan 3 x 32 = 96 split slices against a merge budget of
    64.  The levels of one call now share the budget, so the workspace is that of at most 64 // 3 slices per level."""
    lib = _lib.load()
    B, Hq, D, n = 64, 8, 128, 16384
    rows = B * Hq
    per_slice = rows * D * 4 + rows * 4
    unique = rows * D * 2 + rows * 4  # the unique pass's own 16-bit partial + LSE (two-stream form), behind the levels
    d = _decode(B, Hq, 1, D, [(1, n), (2, n), (4, n)])
    ws = lib.hyd_decode_workspace_bytes(C.byref(d))
    assert unique < ws <= 3 * (64 // 3) * per_slice + unique
    one = _decode(B, Hq, 1, D, [(1, n)])
    assert lib.hyd_decode_workspace_bytes(C.byref(one)) == 32 * per_slice + unique  # a single level keeps its 32 slices


def test_phase_argument_is_validated():
    lib = _lib.load()
    d = _decode(4, 8, 8, 128, [(1, 64)])
    d.phase = 7
    d.suffix.q = d.suffix.out = d.suffix.k = d.suffix.v = 0x10000
    assert lib.hyd_decode_attn_fused(C.byref(d), None) == -1 and "phase" in lib.hyd_last_error_string().decode()


def test_bench_suffix_schedule_is_a_uniform_cover_for_any_step_count():
    import bench
    for steps in (8, 20, 100, 128, 256, 300):
        s = bench.suffix_schedule(steps, 128)
        assert len(s) == steps and min(s) >= 1 and max(s) <= 128
        assert abs(sum(s) / steps - 64.5) < 1.0, (steps, sum(s) / steps)
        assert max(s) - min(s) >= 100
    assert sorted(bench.suffix_schedule(256, 128)) == sorted(list(range(1, 129)) * 2)


def test_product_path_has_no_cpu_fallback():
    import torch
    from hydragen_amd.attention import hydragen_attention_nopad
    q = torch.zeros(2, 1, 4, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hydragen_attention_nopad(q, q, q, [q[:1]], [q[:1]])


def test_product_sources_never_touch_the_oracle():
    for path in list((REPO / "hydragen_amd").rglob("*.py")) + list((REPO / "hydragen_amd").rglob("*.hip")) + \
            list((REPO / "hydragen_amd").rglob("*.h")):
        txt = path.read_text()
        assert "oracle" not in txt.replace("no CPU oracle", "").replace("or the CPU oracle", ""), path


def test_head_dim_padding_rule():
    """Which head dims the Python mirrors accept (hydragen_amd/flash.py): 64 / 128 / 256 natively, other multiples of 8 up
    to 256 (flash-attn's range) padded to the next of the three, everything else refused like an unsupported shape."""
    from hydragen_amd.flash import padded_head_dim

    assert [padded_head_dim(d) for d in (8, 56, 64, 72, 80, 96, 120, 128, 136, 248, 256)] == \
        [64, 64, 64, 128, 128, 128, 128, 128, 256, 256, 256]
    for d in (0, 4, 100, 264, 512):
        with pytest.raises(NotImplementedError):
            padded_head_dim(d)
