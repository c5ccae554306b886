"""-m gpu: edge cases of the operator surface -- maximum hierarchy depth, strided cache views, several
queries per sequence, HIP-graph capture of the bare operator (what the reference's microbenchmark does,
hydragen/benchmark_utils.py:140-170), head_dim 64 through the fused path."""
import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import _round, make_case
from tests.gpu_util import assert_close, assert_close_l2, case_to_device, dev

pytestmark = pytest.mark.gpu


def _rand(rng, shape, dt):
    return _round(rng.standard_normal(shape, dtype=np.float32), dt)


def test_eight_shared_levels():
    from hydragen_amd.attention import hydragen_attention_nopad

    dt, rng = "f16", np.random.default_rng(21)
    B, Hq, Hkv, D = 16, 8, 4, 128
    q = _rand(rng, (B, 1, Hq, D), dt)
    k, v = _rand(rng, (B, 9, Hkv, D), dt), _rand(rng, (B, 9, Hkv, D), dt)
    sbs = [1, 1, 2, 2, 4, 8, 8, 16]
    sks = [_rand(rng, (sb, 5 + 3 * i, Hkv, D), dt) for i, sb in enumerate(sbs)]
    svs = [_rand(rng, (sb, 5 + 3 * i, Hkv, D), dt) for i, sb in enumerate(sbs)]
    lens = np.asarray([9, 1, 4, 7] * 4, dtype=np.int32)
    out = hydragen_attention_nopad(dev(q, dt), dev(k, dt), dev(v, dt), [dev(x, dt) for x in sks], [dev(x, dt) for x in svs], dev(lens))
    torch.cuda.synchronize()
    assert_close(out.float().cpu().numpy(), O.hydragen_attention_nopad(q, k, v, sks, svs, lens), dt, "8 levels")
    # deeper hierarchies than the one-call operator's HYD_MAX_LEVELS run the general form (a prefix pass per level, the
    # suffix kernel with its LSE, N-way merge): the reference loops over any number of levels (attention.py:250-341)
    sbs12 = sbs + [16, 4, 2, 1]
    sks12 = sks + [_rand(rng, (sb, 3 + i, Hkv, D), dt) for i, sb in enumerate(sbs12[8:])]
    svs12 = svs + [_rand(rng, (sb, 3 + i, Hkv, D), dt) for i, sb in enumerate(sbs12[8:])]
    want = O.hydragen_attention_nopad(q, k, v, sks12, svs12, lens)
    out = hydragen_attention_nopad(dev(q, dt), dev(k, dt), dev(v, dt), [dev(x, dt) for x in sks12], [dev(x, dt) for x in svs12], dev(lens))
    torch.cuda.synchronize()
    assert_close(out.float().cpu().numpy(), want, dt, "12 levels")
    # ... also inside a captured graph (no host synchronisation on that path), and with no unique keys at all
    g = torch.cuda.CUDAGraph()
    dq, dk, dv, dl = dev(q, dt), dev(k, dt), dev(v, dt), dev(lens)
    dks, dvs = [dev(x, dt) for x in sks12], [dev(x, dt) for x in svs12]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hydragen_attention_nopad(dq, dk, dv, dks, dvs, dl)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            og = hydragen_attention_nopad(dq, dk, dv, dks, dvs, dl)
    g.replay()
    torch.cuda.synchronize()
    assert_close(og.float().cpu().numpy(), want, dt, "12 levels, graph replay")
    zl = np.zeros_like(lens)
    out0 = hydragen_attention_nopad(dq, dk, dv, dks, dvs, dev(zl))
    torch.cuda.synchronize()
    assert_close(out0.float().cpu().numpy(), O.hydragen_attention_nopad(q, k, v, sks12, svs12, zl), dt, "12 levels, empty suffixes")


def test_merge_of_more_partials_than_one_launch_takes():
    """_combine_many: 70 partials are merged in groups (64 per launch) through (out, merged LSE) pairs."""
    from hydragen_amd.attention import _combine_many

    rng = np.random.default_rng(5)
    n, shape = 70, (3, 2, 4, 64)
    outs = [rng.standard_normal(shape).astype(np.float32) for _ in range(n)]
    lses = [rng.standard_normal(shape[:-1]).astype(np.float32) * 3 for _ in range(n)]
    got = _combine_many([torch.from_numpy(o).cuda() for o in outs], [torch.from_numpy(l).cuda() for l in lses])
    torch.cuda.synchronize()
    L = np.stack(lses).astype(np.float64)
    w = np.exp(L - L.max(0))
    want = (np.stack(outs).astype(np.float64) * w[..., None]).sum(0) / w.sum(0)[..., None]
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_strided_cache_views_and_multi_query(dt):
    """k/v handed over as views of a larger cache (llama.py:259-262 passes k_out[:bs]) and of a fused
    kv buffer: only strides change, no copies.  nq = 2 queries per sequence with seq_lens."""
    from hydragen_amd.attention import hydragen_attention_nopad

    rng = np.random.default_rng(22)
    B, nq, Hq, Hkv, D, S = 6, 2, 8, 2, 64, 20
    q = _rand(rng, (B, nq, Hq, D), dt)
    kv = _rand(rng, (B + 2, S + 12, 2, Hkv, D), dt)          # [maxB, maxS, (k|v), Hkv, D]
    sk, sv = _rand(rng, (2, 37, Hkv, D), dt), _rand(rng, (2, 37, Hkv, D), dt)
    lens = np.asarray([20, 3, 1, 17, 20, 8], dtype=np.int32)
    tkv = dev(kv, dt)
    tk, tv = tkv[:B, :S, 0], tkv[:B, :S, 1]                   # non-contiguous views
    assert not tk.is_contiguous()
    out = hydragen_attention_nopad(dev(q, dt), tk, tv, [dev(sk, dt)], [dev(sv, dt)], dev(lens))
    torch.cuda.synchronize()
    want = O.hydragen_attention_nopad(q, kv[:B, :S, 0], kv[:B, :S, 1], [sk], [sv], lens)
    assert_close(out.float().cpu().numpy(), want, dt, "strided views, nq=2")


def test_operator_under_hip_graph_capture():
    """The library must be capture-safe: no allocation, no sync, current-stream launches only."""
    from hydragen_amd.attention import hydragen_attention_nopad

    dt, rng = "bf16", np.random.default_rng(23)
    B, Hq, Hkv, D = 64, 8, 8, 128
    q = dev(_rand(rng, (B, 1, Hq, D), dt), dt)
    k, v = dev(_rand(rng, (B, 32, Hkv, D), dt), dt), dev(_rand(rng, (B, 32, Hkv, D), dt), dt)
    sk, sv = dev(_rand(rng, (1, 300, Hkv, D), dt), dt), dev(_rand(rng, (1, 300, Hkv, D), dt), dt)
    lens = torch.randint(1, 33, (B,), device="cuda:0", dtype=torch.int64)
    eager = hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    # new data through the same captured graph (static input buffers, as llama.py:818-821)
    q.copy_(torch.randn_like(q))
    lens.copy_(torch.randint(1, 33, (B,), device="cuda:0"))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("causal", [False, True])
def test_prefix_deferred_rescale_is_exact(dt, causal):
    """The prefix kernel raises its running maximum lazily (only when a 32-key block exceeds it by more than
    2^8) and then rescales O and l in a rarely taken branch.  Bounded random data never takes that branch
    after the first block, so spike individual keys late in the sequence (raw q.k far above every other
    score, at several block positions, twice in a row) and compare out AND lse with the float64 oracle."""
    from hydragen_amd.flash import flash_attention

    rng = np.random.default_rng(23)
    b, sq, sk, hq, hkv, D = 1, 160, 1000, 4, 2, 128
    q = _rand(rng, (b, sq, hq, D), dt)
    k, v = _rand(rng, (b, sk, hkv, D), dt), _rand(rng, (b, sk, hkv, D), dt)
    for pos, f in ((70, 5.0), (333, 9.0), (334, 14.0), (640, 20.0), (959, 28.0), (999, 40.0)):
        k[:, pos] = _round(k[:, pos] * f, dt)
    q[:, :, 0] = _round(np.abs(q[:, :, 0]) * np.sign(k[:, 959, 0])[:, None, :], dt)  # head 0: every row aligns with key 959
    out, lse = flash_attention(dev(q, dt), dev(k, dt), dev(v, dt), causal=causal)
    torch.cuda.synchronize()
    want, wlse = O.flash_attention(q, k, v, causal=causal)
    assert_close(out.float().cpu().numpy(), want, dt, "spiked keys: out")
    assert (np.abs(lse.cpu().numpy() - wlse) <= 2e-3 + 1e-5 * np.abs(wlse)).all()


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("kind", ["all_far_below_zero", "ramp_per_block", "ramp_per_key", "plateau_just_under_the_bound", "late_spike_after_plateau"])
def test_prefix_sum_tested_softmax_on_adversarial_scores(dt, kind):
    """The 8-wave prefix unit has no maximum tree in its hot path: a block's probabilities are formed against the reference
    as it stands and the lane's sum of them is tested against 2^8 (prefix_unit_w64.h, LAZY).  Score patterns chosen against
    that test: every score far below zero (the reference must come down from its start value, nothing may underflow),
    maxima that climb by less than the bound per block / per key (sums of moderate terms trip it, the reference follows),
    a plateau of equal scores just under the bound and a spike behind it.  out AND lse against the float64 oracle."""
    from hydragen_amd.flash import flash_attention

    rng = np.random.default_rng({"all_far_below_zero": 1, "ramp_per_block": 2, "ramp_per_key": 3, "plateau_just_under_the_bound": 4,
                                 "late_spike_after_plateau": 5}[kind])
    b, sq, sk, hq, hkv, D = 1, 160, 1000, 4, 2, 128
    unit = np.zeros(D, np.float32)
    unit[:16] = 1.0                                   # q . k = 16 a b for q = a unit, k = b unit; log2-units = 16 a b * D^-0.5 * log2(e)
    per = 16 * D ** -0.5 * 1.4426950408889634          # log2 units per unit of a * b
    q = np.broadcast_to(unit, (b, sq, hq, D)).copy() + 0.01 * rng.standard_normal((b, sq, hq, D)).astype(np.float32)
    j = np.arange(sk, dtype=np.float32)
    if kind == "all_far_below_zero":
        amp = np.full(sk, -60.0 / per) + rng.standard_normal(sk).astype(np.float32) * 0.5      # every score ~ -60 +- 1 (log2 units)
    elif kind == "ramp_per_block":
        amp = (6.0 * (j // 32)) / per                                                             # +6 per 32-key block: 2^6 x 16 keys > 2^8
    elif kind == "ramp_per_key":
        amp = (0.35 * j) / per                                                                    # +11 per block, smooth
    elif kind == "plateau_just_under_the_bound":
        amp = np.where(j < 40, 0.0, 7.5) / per                                                    # from key 40 on: 7.5 above the first reference
    else:
        amp = np.where(j < 40, 0.0, 7.5) / per
        amp[977] = 30.0 / per
    k = (amp[:, None] * unit[None, :])[None, :, None, :].repeat(hkv, 2).astype(np.float32)
    k = k + 0.01 * rng.standard_normal(k.shape).astype(np.float32)
    q, k = _round(q, dt), _round(k, dt)
    v = _rand(rng, (b, sk, hkv, D), dt)
    for causal in (False, True):
        out, lse = flash_attention(dev(q, dt), dev(k, dt), dev(v, dt), causal=causal)
        torch.cuda.synchronize()
        want, wlse = O.flash_attention(q, k, v, causal=causal)
        assert np.isfinite(out.float().cpu().numpy()).all() and np.isfinite(lse.cpu().numpy()).all()
        assert_close(out.float().cpu().numpy(), want, dt, f"{kind} causal={causal}: out")
        assert (np.abs(lse.cpu().numpy() - wlse) <= 2e-3 + 1e-5 * np.abs(wlse)).all(), kind


@pytest.mark.parametrize("sk", [1, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 160, 255, 256, 257, 300, 511, 513])
def test_prefix_lengths_around_block_boundaries(sk):
    """Keys are staged in 32-key blocks (two per 64-key half of a 128-key tile) by bounds-checked DMA: every
    length around those boundaries, D = 128 and 64, against the oracle."""
    from hydragen_amd.flash import flash_attention

    dt, rng = "bf16", np.random.default_rng(100 + sk)
    for D, hq, hkv in ((128, 4, 2), (64, 2, 2)):
        q = _rand(rng, (2, 70, hq, D), dt)
        k, v = _rand(rng, (2, sk, hkv, D), dt), _rand(rng, (2, sk, hkv, D), dt)
        out, lse = flash_attention(dev(q, dt), dev(k, dt), dev(v, dt))
        torch.cuda.synchronize()
        want, wlse = O.flash_attention(q, k, v)
        assert_close(out.float().cpu().numpy(), want, dt, f"sk={sk} D={D}")
        assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("D,Hq,Hkv,nq", [(128, 8, 1, 1), (64, 8, 2, 1), (128, 4, 1, 2), (128, 8, 1, 3), (256, 8, 1, 1), (256, 4, 1, 3)])
def test_grouped_query_suffix_kernel(dt, D, Hq, Hkv, nq):
    """Shapes that the dispatcher sends to the matrix-core suffix kernel (>= 4 query rows per (sequence, kv head),
    >= 256 units): ragged lengths incl. 1, a 32-key boundary and the full cache; 4 / 8 / 24 rows per unit (the last
    one needs two 16-row chunks); through `flash_attention_seqlen` (out + LSE) and through the fused operator with
    one and two shared levels (epilogue merge)."""
    from hydragen_amd.attention import hydragen_attention_nopad
    from hydragen_amd.flash import flash_attention_seqlen

    rng = np.random.default_rng(31 + D + Hq + nq)
    B, S = 256 // Hkv * (2 if Hkv == 2 else 1), 70
    q = _rand(rng, (B, nq, Hq, D), dt)
    k, v = _rand(rng, (B, S, Hkv, D), dt), _rand(rng, (B, S, Hkv, D), dt)
    lens = rng.integers(1, S + 1, B).astype(np.int32)
    lens[:6] = [1, 31, 32, 33, 64, S]
    out, lse = flash_attention_seqlen(dev(q, dt), dev(k, dt), dev(v, dt), seq_len=dev(lens))
    torch.cuda.synchronize()
    want, wlse = O.flash_attention_seqlen(q, k, v, lens)
    assert_close(out.float().cpu().numpy(), want, dt, "gqa suffix: out")
    assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3
    if nq == 1:
        sks = [_rand(rng, (1, 45, Hkv, D), dt), _rand(rng, (4, 9, Hkv, D), dt)]
        svs = [_rand(rng, (1, 45, Hkv, D), dt), _rand(rng, (4, 9, Hkv, D), dt)]
        for n in (1, 2):
            got = hydragen_attention_nopad(dev(q, dt), dev(k, dt), dev(v, dt), [dev(x, dt) for x in sks[:n]],
                                           [dev(x, dt) for x in svs[:n]], dev(lens))
            torch.cuda.synchronize()
            assert_close(got.float().cpu().numpy(), O.hydragen_attention_nopad(q, k, v, sks[:n], svs[:n], lens), dt,
                         f"gqa suffix fused, {n} level(s)")


def test_repeated_launches_are_bit_identical():
    """No atomics anywhere on the path, so every launch must reproduce the same bits; a difference would mean a
    race (LDS ring slot reused too early, a DMA not waited for, a missing barrier) that a tolerance check can miss.
    Shapes cover the pipelined prefix kernel (full tiles, a ragged tail, split-KV), the dot-product suffix kernel
    and the matrix-core suffix kernel, back to back so that launches overlap their tails."""
    from hydragen_amd.attention import hydragen_attention_nopad

    g = torch.Generator(device="cuda:0").manual_seed(5)
    r = lambda *s: torch.randn(*s, device="cuda:0", dtype=torch.bfloat16, generator=g)
    for B, P, S, Hq, Hkv in ((512, 2048, 33, 8, 8), (256, 1000, 70, 8, 1), (16, 4097, 5, 16, 4), (1024, 300, 129, 4, 4)):
        q, k, v, sk, sv = r(B, 1, Hq, 128), r(B, S, Hkv, 128), r(B, S, Hkv, 128), r(1, P, Hkv, 128), r(1, P, Hkv, 128)
        lens = torch.randint(1, S + 1, (B,), device="cuda:0", generator=g, dtype=torch.int32)
        outs = [hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens) for _ in range(40)]
        torch.cuda.synchronize()
        for o in outs[1:]:
            assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16)), (B, P, S, Hq, Hkv)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("sk", [70, 300])
def test_prefix_256_row_workgroups(dt, sk):
    """Enough query rows (8 kv heads x 8229 rows = 520 128-row blocks >= 2 rounds of the chip) for the planner to pick
    the 256-row instantiation of the prefix kernel; the last workgroup of every head is ragged (8229 = 32 * 256 + 37)."""
    from hydragen_amd import _lib
    from hydragen_amd.flash import flash_attention

    rng = np.random.default_rng(41 + sk)
    q = _rand(rng, (1, 8229, 8, 128), dt)
    k, v = _rand(rng, (1, sk, 8, 128), dt), _rand(rng, (1, sk, 8, 128), dt)
    out, lse = flash_attention(dev(q, dt), dev(k, dt), dev(v, dt))
    torch.cuda.synchronize()
    want, wlse = O.flash_attention(q, k, v)
    assert_close(out.float().cpu().numpy(), want, dt, "256-row workgroups: out")
    assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("hq,hkv,nq", [(8, 8, 1), (8, 2, 1), (8, 1, 1), (4, 4, 3)])
def test_padding_beyond_seq_len_is_never_read_into_the_result(dt, hq, hkv, nq):
    """The unique cache is handed over whole with seq_lens (llama.py:259-262,569): whatever sits behind a sequence's
    length -- here NaN and Inf -- must not reach the output (0 * NaN would), for the dot-product and the matrix-core
    suffix kernels alike, with lengths that end inside a 32-key step."""
    from hydragen_amd.flash import flash_attention_seqlen

    rng = np.random.default_rng(31 + hq + hkv + nq)
    B, S, D = 5, 70, 128
    lens = np.asarray([1, 33, 70, 17, 64], dtype=np.int32)
    q, k, v = _rand(rng, (B, nq, hq, D), dt), _rand(rng, (B, S, hkv, D), dt), _rand(rng, (B, S, hkv, D), dt)
    kp, vp = k.copy(), v.copy()
    for b in range(B):
        kp[b, lens[b]:] = np.nan
        vp[b, lens[b]:] = np.inf if b % 2 else np.nan
    out, lse = flash_attention_seqlen(dev(q, dt), dev(kp, dt), dev(vp, dt), seq_len=dev(lens))
    torch.cuda.synchronize()
    want, wlse = O.flash_attention_seqlen(q, k, v, lens)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all(), "padding leaked into the output"
    assert_close(got, want, dt, "poisoned padding")
    assert np.abs(lse.cpu().numpy() - wlse).max() < 2e-3


def test_cached_parameter_blocks_follow_the_callers_tensors():
    """The eager host path caches marshalled parameter blocks per (addresses, shapes, strides): repeated calls must keep
    reading the CURRENT contents of the caller's tensors (seq_lens changed in place, K/V overwritten), and calls with
    other shapes must not collide."""
    from hydragen_amd.attention import hydragen_attention_nopad

    torch.manual_seed(41)  # k / q are redrawn in place below from torch's global generator
    dt, rng = "bf16", np.random.default_rng(41)
    B, Hq, Hkv, D, S, P = 6, 8, 8, 128, 24, 150
    q = dev(_rand(rng, (B, 1, Hq, D), dt), dt)
    k, v = dev(_rand(rng, (B, S, Hkv, D), dt), dt), dev(_rand(rng, (B, S, Hkv, D), dt), dt)
    sk, sv = dev(_rand(rng, (1, P, Hkv, D), dt), dt), dev(_rand(rng, (1, P, Hkv, D), dt), dt)
    lens = torch.full((B,), 3, dtype=torch.int32, device="cuda:0")
    f = lambda t: t.float().cpu().numpy()
    for step in range(4):
        out = hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens)
        torch.cuda.synchronize()
        want = O.hydragen_attention_nopad(f(q), f(k), f(v), [f(sk)], [f(sv)], lens.cpu().numpy())
        assert_close(f(out), want, dt, f"cached call {step}")
        lens += 5                       # in place: same tensor, new contents
        k[:, step].normal_()            # the cache keeps growing between decode steps
        q.normal_()
    out2 = hydragen_attention_nopad(q[:4], k[:4], v[:4], [sk], [sv], seq_len=lens[:4])
    torch.cuda.synchronize()
    want2 = O.hydragen_attention_nopad(f(q[:4]), f(k[:4]), f(v[:4]), [f(sk)], [f(sv)], lens[:4].cpu().numpy())
    assert_close(f(out2), want2, dt, "other shapes")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("sizes,qh,kvh,dim", [
    ([[64], [8, 8, 8, 8]], 4, 4, 64),                 # BASELINE config 1 literal
    ([[3], [6, 7]], 8, 1, 128),                       # ragged unique lengths, 8 query heads on one kv head
    ([[130, 130], [1, 200, 31, 2, 17, 150]], 8, 2, 128),  # two groups, 4 waves per unit (>= 128 keys allocated, few units)
    ([[33], [1]], 8, 8, 64),                          # one sequence
    ([[1000], [5, 9]], 16, 4, 128),                   # the longest prefix that still counts as a small level
])
def test_single_launch_form_of_tiny_problems(sizes, qh, kvh, dim, dtype):
    """hyd_decode_params.single_launch_small: problems that are launch latency run as ONE kernel (the grouped-query kernel
    walks the group's shared keys, then the sequence's own).  Against the float64 oracle with the operator's usual bounds,
    and against the two-pass form of the same call (which rounds a partial in between)."""
    import ctypes as C
    from hydragen_amd import attention as A, _lib
    from oracle import hydragen_oracle as O

    case = make_case(sizes=sizes, qheads=qh, kvheads=kvh, dim=dim, dtype=dtype, seed=4242, force_seq_lens=True)
    d = case_to_device(case)
    want = O.hydragen_attention(case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"], case["shared_cu_seq_lens"],
                                case["shared_max_seq_lens"], case["use_varlens"], case["seq_lens"])
    seen = []
    orig = A._launch_decode

    def spy(lib, p, two_stream, stream, single=1):
        p.phase, p.shared_max_workgroups, p.single_launch_small = _lib.HYD_PHASE_ALL, 0, single
        _lib.check(lib.hyd_decode_attn_fused(C.byref(p), stream))
        seen.append(single)

    outs = {}
    try:
        for single in (1, 0):
            A._PARAM_CACHE.clear()
            A._launch_decode = lambda lib, p, ts, st, single=single: spy(lib, p, ts, st, single)
            outs[single] = A.hydragen_attention(**d).float().cpu().numpy()
    finally:
        A._launch_decode = orig
        A._PARAM_CACHE.clear()
    assert seen == [1, 0]
    assert_close_l2(outs[1], want, dtype, "single launch vs float64 oracle")
    assert_close_l2(outs[0], want, dtype, "two passes vs float64 oracle")
    if dtype == "bf16":  # the two-pass form rounds a bf16 partial in between: equal bits would mean the flag was ignored
        assert not np.array_equal(outs[0], outs[1])
