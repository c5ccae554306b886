import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(GOLDEN / f"{name}.npz")
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d["meta"]).decode())
    return d


@pytest.fixture(scope="session")
def oracle_so():
    """Build (if needed) and load the plain-C oracle.  Test infrastructure only."""
    import ctypes
    import subprocess

    src = REPO / "oracle" / "hydragen_oracle.c"
    so = REPO / "oracle" / "libhydragen_oracle.so"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", str(src), "-o", str(so), "-lm"])
    return ctypes.CDLL(str(so))
