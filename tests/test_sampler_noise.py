"""-m "not gpu": the token sampler's uniform variate stays strictly inside (0, 1) for every 32-bit Philox word, so the
Gumbel noise -ln(-ln u) of hydragen_amd/csrc/layer_ops.hip is finite.  (Round 3's form, (bits >> 8) 2^-24 + 2^-25, rounded
to exactly 1.0 for the largest word: -ln u = 0, noise +inf, and that column won the argmax whatever its logit.)"""
import re
from pathlib import Path

import numpy as np

SRC = Path(__file__).resolve().parent.parent / "hydragen_amd" / "csrc" / "layer_ops.hip"


def _u(bits: np.ndarray) -> np.ndarray:
    """fp32 evaluation of the kernel's expression, operation by operation"""
    k = (bits >> np.uint32(9)).astype(np.float32)          # exact: k < 2^23
    return (k + np.float32(0.5)) * np.float32(2.0 ** -23)   # k + 1/2 has at most 24 significant bits: exact


def test_uniform_variate_is_strictly_inside_the_unit_interval():
    text = SRC.read_text()
    assert re.search(r"const float u = \(\(float\)\(bits >> 9\) \+ 0\.5f\) \* 0x1p-23f;", text), "gumbel() changed: update this mirror"
    # -ln u is clamped from below, so the noise is finite whatever the hardware's approximate log2 returns near u = 1
    assert re.search(r"const float e = fmaxf\(-kLn2 \* fast_log2\(u\), 0x1p-25f\);", text), "gumbel() lost its clamp"
    for sloppy_log in (0.0, -8.6e-8):  # log2(u) at the largest u: a sloppy unit may return exactly 0
        e = max(-np.log(2.0) * sloppy_log, 2.0 ** -25)
        assert np.isfinite(-np.log(e)) and -np.log(e) <= 17.4
    edge = np.array([0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFE00, 0xFFFFFFFF], dtype=np.uint32)
    rnd = np.random.default_rng(0).integers(0, 2 ** 32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    for bits in (edge, rnd):
        u = _u(bits)
        assert u.dtype == np.float32 and (u > 0).all() and (u < 1).all()
        g = -np.log(-np.log(u.astype(np.float64)))
        assert np.isfinite(g).all()
    # the form this replaces did reach 1.0 (a tie that rounds to even)
    old = (np.uint32(0xFFFFFFFF) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) + np.float32(2.0 ** -25)
    assert old == np.float32(1.0)
    # every value is hit by exactly 2^9 words and the grid is uniform: mean 1/2
    assert abs(float(_u(rnd).mean()) - 0.5) < 2e-3
