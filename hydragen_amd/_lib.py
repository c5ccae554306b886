"""
ctypes binding of libhydragen_hip.so (C ABI: include/hydragen_hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every
operator raises `HydragenLibraryError` -- it never routes through torch or the CPU oracle.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

HYD_MAX_LEVELS = 8
HYD_F16, HYD_BF16, HYD_F32 = 0, 1, 2
HYD_LSE_BQH, HYD_LSE_BHQ = 0, 1
HYD_PHASE_ALL, HYD_PHASE_SHARED, HYD_PHASE_UNIQUE, HYD_PHASE_UNIQUE_PARTIAL, HYD_PHASE_MERGE = 0, 1, 2, 3, 4

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libhydragen_hip.so"


class HydragenLibraryError(RuntimeError):
    pass


class PrefixParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p), ("lse", C.c_void_p),
        ("cu_seqlens_k", C.c_void_p), ("cu_seqlens_q", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("k_group_stride", C.c_int64), ("k_tok_stride", C.c_int64), ("k_head_stride", C.c_int64),
        ("v_group_stride", C.c_int64), ("v_tok_stride", C.c_int64), ("v_head_stride", C.c_int64),
        ("dtype", C.c_int32), ("B", C.c_int32), ("nq", C.c_int32), ("Hq", C.c_int32), ("Hkv", C.c_int32),
        ("D", C.c_int32), ("sb", C.c_int32), ("kv_len", C.c_int32), ("max_q_len", C.c_int32),
        ("causal", C.c_int32), ("lse_layout", C.c_int32), ("num_splits", C.c_int32),
        ("softmax_scale", C.c_float), ("reserved_", C.c_int32),
    ]


class Partial(C.Structure):
    _fields_ = [("out", C.c_void_p), ("lse", C.c_void_p), ("count", C.c_int32), ("is_f32", C.c_int32)]


class SuffixParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p), ("lse", C.c_void_p),
        ("seq_lens_i32", C.c_void_p), ("seq_lens_i64", C.c_void_p), ("seq_order", C.c_void_p),
        ("k_batch_stride", C.c_int64), ("k_tok_stride", C.c_int64), ("k_head_stride", C.c_int64),
        ("v_batch_stride", C.c_int64), ("v_tok_stride", C.c_int64), ("v_head_stride", C.c_int64),
        ("dtype", C.c_int32), ("B", C.c_int32), ("nq", C.c_int32), ("Hq", C.c_int32), ("Hkv", C.c_int32),
        ("D", C.c_int32), ("kv_len", C.c_int32), ("n_partials", C.c_int32),
        ("softmax_scale", C.c_float), ("reserved_", C.c_int32),
        ("partials", Partial * HYD_MAX_LEVELS),
    ]


class Level(C.Structure):
    _fields_ = [
        ("k", C.c_void_p), ("v", C.c_void_p), ("cu_seqlens_k", C.c_void_p),
        ("k_group_stride", C.c_int64), ("k_tok_stride", C.c_int64), ("k_head_stride", C.c_int64),
        ("v_group_stride", C.c_int64), ("v_tok_stride", C.c_int64), ("v_head_stride", C.c_int64),
        ("sb", C.c_int32), ("kv_len", C.c_int32),
    ]


class DecodeParams(C.Structure):
    _fields_ = [
        ("suffix", SuffixParams), ("levels", Level * HYD_MAX_LEVELS),
        ("n_levels", C.c_int32), ("phase", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("shared_max_workgroups", C.c_int32), ("f32_partials", C.c_int32),
        ("single_launch_small", C.c_int32), ("reserved", C.c_int32),
    ]


class RopeParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("q_out", C.c_void_p),
        ("k_cache", C.c_void_p), ("v_cache", C.c_void_p), ("cos", C.c_void_p), ("sin", C.c_void_p),
        ("position_ids", C.c_void_p), ("shared_len", C.c_void_p), ("seq_lens", C.c_void_p),
        ("q_batch_stride", C.c_int64), ("k_batch_stride", C.c_int64), ("v_batch_stride", C.c_int64),
        ("kc_batch_stride", C.c_int64), ("kc_tok_stride", C.c_int64), ("kc_head_stride", C.c_int64),
        ("vc_batch_stride", C.c_int64), ("vc_tok_stride", C.c_int64), ("vc_head_stride", C.c_int64),
        ("pos_stride", C.c_int64), ("cs_stride", C.c_int64),
        ("dtype", C.c_int32), ("B", C.c_int32), ("Hq", C.c_int32), ("Hkv", C.c_int32), ("D", C.c_int32),
        ("cache_len", C.c_int32), ("max_pos", C.c_int32), ("reserved", C.c_int32),
    ]


class AddRmsnormParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("residual", C.c_void_p), ("weight", C.c_void_p), ("sum_out", C.c_void_p), ("norm_out", C.c_void_p),
        ("x_row_stride", C.c_int64), ("residual_row_stride", C.c_int64), ("sum_row_stride", C.c_int64),
        ("norm_row_stride", C.c_int64), ("rows", C.c_int64), ("n", C.c_int32), ("dtype", C.c_int32), ("eps", C.c_float),
        ("reserved", C.c_int32),
    ]


class SwigluParams(C.Structure):
    _fields_ = [
        ("gate", C.c_void_p), ("up", C.c_void_p), ("out", C.c_void_p),
        ("gate_row_stride", C.c_int64), ("up_row_stride", C.c_int64), ("out_row_stride", C.c_int64),
        ("rows", C.c_int64), ("n", C.c_int32), ("dtype", C.c_int32),
    ]


class SampleParams(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p), ("out", C.c_void_p), ("row_stride", C.c_int64), ("seed", C.c_uint64), ("offset", C.c_uint64),
        ("rows", C.c_int32), ("n", C.c_int32), ("dtype", C.c_int32), ("temperature", C.c_float),
    ]


class AllReduceParams(C.Structure):
    _fields_ = [
        ("blocks", C.POINTER(C.c_void_p)), ("in_", C.c_void_p), ("out", C.c_void_p), ("count", C.c_int64),
        ("max_bytes", C.c_size_t), ("dtype", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
        ("timeout_log2_polls", C.c_int32),
    ]


# every symbol include/hydragen_hip.h declares
EXPORTS = {
    "hyd_version": (C.c_int, []),
    "hyd_last_error_string": (C.c_char_p, []),
    "hyd_prefix_plan": (C.c_int, [C.POINTER(PrefixParams), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hyd_prefix_workspace_bytes": (C.c_size_t, [C.POINTER(PrefixParams)]),
    "hyd_prefix_attn_fwd": (C.c_int, [C.POINTER(PrefixParams), C.c_void_p]),
    "hyd_suffix_attn_fwd": (C.c_int, [C.POINTER(SuffixParams), C.c_void_p]),
    "hyd_combine_lse": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hyd_decode_workspace_bytes": (C.c_size_t, [C.POINTER(DecodeParams)]),
    "hyd_decode_attn_fused": (C.c_int, [C.POINTER(DecodeParams), C.c_void_p]),
    "hyd_decode_two_stream_ok": (C.c_int, [C.POINTER(DecodeParams)]),
    "hyd_rope_append_decode": (C.c_int, [C.POINTER(RopeParams), C.c_void_p]),
    "hyd_add_rmsnorm": (C.c_int, [C.POINTER(AddRmsnormParams), C.c_void_p]),
    "hyd_swiglu": (C.c_int, [C.POINTER(SwigluParams), C.c_void_p]),
    "hyd_sample_tokens": (C.c_int, [C.POINTER(SampleParams), C.c_void_p]),
    "hyd_ipc_get_handle": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hyd_ipc_open_handle": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "hyd_ipc_close_handle": (C.c_int, [C.c_void_p]),
    "hyd_allreduce_block_bytes": (C.c_size_t, [C.c_int32, C.c_size_t]),
    "hyd_allreduce_sum": (C.c_int, [C.POINTER(AllReduceParams), C.c_void_p]),
    "hyd_allreduce_status": (C.c_void_p, [C.c_void_p]),
    "hyd_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

_lib = None
ABI_VERSION = 500  # HYD_VERSION of include/hydragen_hip.h these mirrors were written against


def lib_path() -> Path:
    return Path(os.environ.get("HYDRAGEN_HIP_LIB", str(_LIB_PATH)))


def load():
    """Load the library (once).  torch is imported first so that the HIP runtime already mapped by
    PyTorch-ROCm (same SONAME libamdhip64.so.7) is the one our kernels launch on."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps torch's libamdhip64 before ours resolves it)

    p = lib_path()
    if not p.exists():
        raise HydragenLibraryError(
            f"{p} not found: build it with `python hydragen_amd/csrc/build.py` "
            "(or __graft_entry__.build()); there is no non-HIP fallback"
        )
    if p == _LIB_PATH:
        # the in-tree library: built by csrc/build.py, which stamps the prefix kernels' register-ownership check (csrc/regcheck.py)
        missing = [s for s in ("prefix_attn_w64.regcheck", "prefix_attn_w64_f16.regcheck") if not (p.parent / s).exists()]
        if missing:
            raise HydragenLibraryError(
                f"{p} was not built by hydragen_amd/csrc/build.py ({', '.join(missing)} missing): the prefix kernels' literal "
                "registers are only safe with a compiler that passed csrc/regcheck.py -- rebuild with build.py")
    try:
        lib = C.CDLL(str(p))
    except OSError as e:  # pragma: no cover
        raise HydragenLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in EXPORTS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HydragenLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.hyd_version()
    if got != ABI_VERSION:  # the struct mirrors above are those of exactly this header version
        raise HydragenLibraryError(
            f"{p} is ABI {got // 100}.{got // 10 % 10}.{got % 10}, the Python mirrors are "
            f"{ABI_VERSION // 100}.{ABI_VERSION // 10 % 10}.{ABI_VERSION % 10}: rebuild it (python hydragen_amd/csrc/build.py --force)")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().hyd_last_error_string().decode()
        if rc == -1:
            raise ValueError(f"hydragen_hip: {msg}")
        if rc == -2:
            raise NotImplementedError(f"hydragen_hip: {msg}")
        raise RuntimeError(f"hydragen_hip error {rc}: {msg}")
