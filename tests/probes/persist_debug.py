"""Debug: the spec4 (P = 16384, B = 3) fixture through the phases of the operator on ONE stream, for several prefix
workgroup limits -- separates the persistent prefix kernel from stream concurrency."""
import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from tests.cases import golden_case_list, make_case
from tests.gpu_util import case_to_device
from tests.conftest import load_golden
from hydragen_amd import attention as A, _lib

name = sys.argv[1] if len(sys.argv) > 1 else "ref_spec4_kv8_f16"
kw = dict(golden_case_list())[name]
case = make_case(**kw)
g = load_golden(name)
d = case_to_device(case)
want = g["out_exact"]
lib = _lib.load()
orig = A._launch_decode

def run(mwg, phases, reps=3):
    errs = []
    def launch(lib_, p, two_stream, stream):
        p.shared_max_workgroups = mwg
        for ph in phases:
            p.phase = ph
            _lib.check(lib_.hyd_decode_attn_fused(C.byref(p), stream))
    A._launch_decode = launch
    A._PARAM_CACHE.clear()
    for _ in range(reps):
        out = A.hydragen_attention(**d)
        torch.cuda.synchronize()
        errs.append(float(np.abs(out.float().cpu().numpy() - want).max()))
    A._launch_decode = orig
    return errs

ALL, SH, UN, UP, MG = _lib.HYD_PHASE_ALL, _lib.HYD_PHASE_SHARED, _lib.HYD_PHASE_UNIQUE, _lib.HYD_PHASE_UNIQUE_PARTIAL, _lib.HYD_PHASE_MERGE
for mwg in (0, 256, 128, 64, 7, 3, 1):
    print(f"max workgroups {mwg:4d}: one call {run(mwg, [ALL])}  shared+unique {run(mwg, [SH, UN])}  shared+partial+merge {run(mwg, [SH, UP, MG])}")
