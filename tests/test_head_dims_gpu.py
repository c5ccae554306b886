"""-m gpu: head dims other than the kernels' own 64 / 128 / 256.  flash-attn, which the reference calls with whatever
head_dim the model has (/root/reference/hydragen/flash.py:295-304), accepts any multiple of 8 up to 256; here 256 is a
native instantiation (one query block per wave in the prefix pass, 32 lanes per key in the suffix pass) and the other
multiples of 8 run zero-padded to 64 / 128 / 256 with the TRUE head dim's softmax scale (hyd_*_params.softmax_scale).
Checked against the float64 oracle, which knows nothing of the padding."""
import zlib

import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import _round, make_case
from tests.gpu_util import assert_close, case_to_device, dev

pytestmark = pytest.mark.gpu


def _rand(rng, shape, dt):
    return _round(rng.standard_normal(shape, dtype=np.float32), dt)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("D", [8, 32, 80, 96, 120, 136, 192, 256])
def test_primitives_with_other_head_dims(dt, D):
    from hydragen_amd.flash import flash_attention, flash_attention_seqlen

    rng = np.random.default_rng(zlib.crc32(repr((dt, D)).encode()))
    b, sq, sk, hq, hkv = 2, 7, 150, 8, 2
    q, k, v = _rand(rng, (b, sq, hq, D), dt), _rand(rng, (b, sk, hkv, D), dt), _rand(rng, (b, sk, hkv, D), dt)
    for causal in (False, True):
        out, lse = flash_attention(dev(q, dt), dev(k, dt), dev(v, dt), causal=causal)
        want, wlse = O.flash_attention(q, k, v, causal=causal)
        assert out.shape == (b, sq, hq, D) and out.is_contiguous()
        assert_close(out.float().cpu().numpy(), want, dt, f"flash_attention D={D} causal={causal}")
        assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3
    q1 = _rand(rng, (3, 1, hq, D), dt)
    k1, v1 = _rand(rng, (3, 40, hkv, D), dt), _rand(rng, (3, 40, hkv, D), dt)
    sl = np.asarray([40, 1, 17], dtype=np.int32)
    out, lse = flash_attention_seqlen(dev(q1, dt), dev(k1, dt), dev(v1, dt), seq_len=dev(sl))
    want, wlse = O.flash_attention_seqlen(q1, k1, v1, sl)
    assert_close(out.float().cpu().numpy(), want, dt, f"seqlen D={D}")
    assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("D", [80, 96, 160, 256])
@pytest.mark.parametrize("sizes", [[[64], [8, 8, 8, 8]], [[48], [9, 10], [5, 2, 3, 4]]])
def test_hydragen_attention_with_other_head_dims(dt, D, sizes):
    from hydragen_amd.attention import hydragen_attention

    case = make_case(sizes=sizes, qheads=8, kvheads=2, dim=D, dtype=dt, seed=D + len(sizes), force_seq_lens=True)
    out = hydragen_attention(**case_to_device(case))
    torch.cuda.synchronize()
    want = O.hydragen_attention(case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"],
                                case["shared_cu_seq_lens"], case["shared_max_seq_lens"], case["use_varlens"], case["seq_lens"])
    assert tuple(out.shape) == case["q"].shape
    assert_close(out.float().cpu().numpy(), want, dt, f"hydragen_attention D={D}")


def test_unsupported_head_dims_raise():
    from hydragen_amd.flash import flash_attention

    for D in (264, 100, 512):
        q = torch.randn(1, 2, 4, D, device="cuda", dtype=torch.float16)
        with pytest.raises(NotImplementedError):
            flash_attention(q, q, q)


def test_softmax_scale_field_of_the_c_abi():
    """An explicit scale through the C ABI == the default one when it equals D^-0.5, and is honoured otherwise
    (checked on the LSE, which carries the scale)."""
    import hydragen_amd.flash as F

    rng = np.random.default_rng(5)
    q, k, v = _rand(rng, (2, 3, 4, 64), "f16"), _rand(rng, (2, 50, 4, 64), "f16"), _rand(rng, (2, 50, 4, 64), "f16")
    o0, l0 = F.flash_attention(dev(q, "f16"), dev(k, "f16"), dev(v, "f16"))
    with F.true_head_dim_scale(64):
        o1, l1 = F.flash_attention(dev(q, "f16"), dev(k, "f16"), dev(v, "f16"))
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    with F.true_head_dim_scale(16):  # scale 0.25 instead of 0.125
        o2, l2 = F.flash_attention(dev(q, "f16"), dev(k, "f16"), dev(v, "f16"))
    want, wlse = O.flash_attention(q * 2.0, k, v)  # doubling q doubles the scale
    assert_close(o2.float().cpu().numpy(), want, "f16", "explicit scale")
    assert np.abs(l2.cpu().numpy() - wlse).max() < 2e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("causal", [False, True])
def test_head_dim_256_several_row_blocks_and_key_tiles(dt, causal):
    """D = 256 natively: more than one 128-row block per kv head, several 32-key ring rounds with a ragged tail, grouped
    queries, dense and bottom-right causal; and a decode step (shared prefix + ragged unique suffix, 8 q heads per kv head
    -> the dot-product suffix kernel with 32 lanes per key and row chunks)."""
    from hydragen_amd.attention import hydragen_attention_nopad
    from hydragen_amd.flash import flash_attention

    rng = np.random.default_rng(256 + causal)
    b, sq, sk, hq, hkv, D = 1, 300, 700, 4, 2, 256
    q, k, v = _rand(rng, (b, sq, hq, D), dt), _rand(rng, (b, sk, hkv, D), dt), _rand(rng, (b, sk, hkv, D), dt)
    out, lse = flash_attention(dev(q, dt), dev(k, dt), dev(v, dt), causal=causal)
    want, wlse = O.flash_attention(q, k, v, causal=causal)
    assert_close(out.float().cpu().numpy(), want, dt, f"flash_attention D=256 causal={causal}")
    assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3
    if causal:
        return
    B, P, S, Hq, Hkv = 48, 333, 40, 8, 1
    q1, k1, v1 = _rand(rng, (B, 1, Hq, D), dt), _rand(rng, (B, S, Hkv, D), dt), _rand(rng, (B, S, Hkv, D), dt)
    sk1, sv1 = _rand(rng, (1, P, Hkv, D), dt), _rand(rng, (1, P, Hkv, D), dt)
    sl = rng.integers(0, S + 1, B).astype(np.int32)
    got = hydragen_attention_nopad(dev(q1, dt), dev(k1, dt), dev(v1, dt), [dev(sk1, dt)], [dev(sv1, dt)], dev(sl))
    torch.cuda.synchronize()
    want = O.hydragen_attention(q1, k1, v1, [sk1], [sv1], [None], [None], [False], sl)
    assert_close(got.float().cpu().numpy(), want, dt, "hydragen_attention D=256 decode")


@pytest.mark.parametrize("D", [64, 72, 88, 104, 120, 128, 136, 200, 256])
def test_empty_split_of_a_short_ragged_group_for_any_softmax_scale(D):
    """A ragged level whose second group is shorter than one split: that group's later split-KV units see no key at all.
    Their state (m = the finite 'minus infinity', l = 0) must drop out of the merge exactly.  It did not for softmax
    scales whose product with the initial maximum rounds upwards (an fma folded the product into `m - m`: exp2 of the
    rounding error = inf, inf * 0 = NaN); found with head_dim 256, so the scale is swept here through the head dim."""
    from hydragen_amd.attention import hydragen_attention

    for dt in ("f16", "bf16"):
        case = make_case(sizes=[[445], [652, 10], [1, 3, 5, 2, 5, 3]], qheads=8, kvheads=1, dim=D, dtype=dt, seed=D,
                         force_seq_lens=True)
        out = hydragen_attention(**case_to_device(case))
        torch.cuda.synchronize()
        want = O.hydragen_attention(case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"],
                                    case["shared_cu_seq_lens"], case["shared_max_seq_lens"], case["use_varlens"], case["seq_lens"])
        assert_close(out.float().cpu().numpy(), want, dt, f"empty split D={D} {dt}")
