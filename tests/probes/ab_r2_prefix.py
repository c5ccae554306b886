#!/usr/bin/env python3
"""A/B of the prefix pass against the round-2 library (build_probe/libhyd_r2.so, built from commit 7aae54c):
    HYDRAGEN_HIP_LIB=build_probe/libhyd_r2.so python tests/probes/ab_r2_prefix.py   (old)   /   without the variable (new)"""
import os, runpy, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hydragen_amd import _lib
if "r2" in os.environ.get("HYDRAGEN_HIP_LIB", ""):
    _lib.ABI_VERSION = 201
    _lib.EXPORTS.pop("hyd_decode_two_stream_ok")
sys.argv = ["kbench.py"] + (sys.argv[1:] if len(sys.argv) > 1 and sys.argv[1] in ("prefix", "suffix", "fused") else ["prefix"] + sys.argv[1:])
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools", "kbench.py"), run_name="__main__")
