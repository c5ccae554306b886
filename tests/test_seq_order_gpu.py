"""-m gpu: the schedule hint for ragged unique lengths (`hyd_suffix_params.seq_order`, `hydragen_amd.flash.seq_order`): the order in
which the suffix kernels hand sequences to the chip.  Only the schedule may depend on it -- every kernel that takes it (the token-row
kernel, the one-unit-per-wave kernel, the grouped-query kernel, the one-call decode operator) must give BIT-identical results for any
permutation -- and the model shell derives it once per generation from the unique prompts' lengths."""
import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import _round
from tests.gpu_util import assert_close, dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _perms(B, seed):
    g = torch.Generator().manual_seed(seed)
    rnd = torch.randperm(B, generator=g).to(torch.int32)
    return {"random": rnd.to(DEV), "reversed": torch.arange(B - 1, -1, -1, dtype=torch.int32, device=DEV),
            "identity": torch.arange(B, dtype=torch.int32, device=DEV)}


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("B,Hq,Hkv,cap,what", [
    (130, 16, 16, 40, "token-row kernel, 4 waves per sequence"),
    (520, 4, 4, 40, "token-row kernel, 2 waves share a sequence"),
    (37, 8, 8, 300, "one-unit-per-wave kernel"),
    (50, 4, 4, 20, "one-unit-per-wave kernel, short (lane-group path)"),
    (90, 32, 8, 70, "grouped-query kernel, 4 kv heads per workgroup"),
    (9, 8, 1, 300, "grouped-query kernel, 4 waves per unit"),
])
def test_suffix_pass_is_bit_identical_under_any_order(dt, B, Hq, Hkv, cap, what):
    from hydragen_amd.flash import flash_attention_seqlen, longest_first, seq_order

    D = 128
    rng = np.random.default_rng(B * 7 + Hq + cap)
    rnd = lambda *s: _round(rng.standard_normal(s, dtype=np.float32), dt)
    q, k, v = rnd(B, 1, Hq, D), rnd(B, cap, Hkv, D), rnd(B, cap, Hkv, D)
    sl = rng.integers(0, cap + 1, B).astype(np.int32)
    sl[:3] = [0, 1, cap]
    tq, tk, tv, tsl = dev(q, dt), dev(k, dt), dev(v, dt), dev(sl)
    out0, lse0 = flash_attention_seqlen(tq, tk, tv, seq_len=tsl)
    want, _ = O.flash_attention_seqlen(q, k, v, sl)
    assert_close(out0.float().cpu().numpy(), want, dt, what)
    perms = _perms(B, B + cap)
    perms["longest first"] = longest_first(tsl)
    assert torch.equal(tsl[perms["longest first"].long()], tsl.sort(descending=True).values)
    for name, perm in perms.items():
        with seq_order(perm):
            out, lse = flash_attention_seqlen(tq, tk, tv, seq_len=tsl)
        assert torch.equal(out, out0) and torch.equal(lse, lse0), (what, name)


def test_seq_order_is_checked_on_entry():
    from hydragen_amd.flash import flash_attention_seqlen, seq_order

    with pytest.raises(ValueError, match="not a permutation"):
        seq_order(torch.tensor([0, 2, 2, 1], dtype=torch.int32, device=DEV))
    with pytest.raises(ValueError, match="int32"):
        seq_order(torch.arange(4, device=DEV))
    q = torch.randn(5, 1, 4, 128, device=DEV, dtype=torch.bfloat16)
    k = torch.randn(5, 8, 4, 128, device=DEV, dtype=torch.bfloat16)
    with seq_order(torch.arange(4, dtype=torch.int32, device=DEV)):
        with pytest.raises(ValueError, match="batch of 5"):
            flash_attention_seqlen(q, k, k, seq_len=torch.full((5,), 8, dtype=torch.int32, device=DEV))


def test_decode_operator_under_an_order_and_its_parameter_cache():
    """hydragen_attention (one C call: prefix pass + suffix pass with the merge) with and without a schedule; the cached parameter
    block is keyed by the order, so switching it on, changing it and switching it off all take effect."""
    from hydragen_amd.attention import hydragen_attention_nopad
    from hydragen_amd.flash import seq_order

    B, H, D, P, cap = 140, 16, 128, 200, 48
    g = torch.Generator(device=DEV).manual_seed(5)
    r = lambda *s: torch.randn(*s, device=DEV, dtype=torch.bfloat16, generator=g)
    q, k, v, sk, sv = r(B, 1, H, D), r(B, cap, H, D), r(B, cap, H, D), r(1, P, H, D), r(1, P, H, D)
    lens = torch.randint(0, cap + 1, (B,), device=DEV, generator=g, dtype=torch.int32)
    out0 = hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens)
    for name, perm in _perms(B, 3).items():
        with seq_order(perm):
            a = hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens)
            b = hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens)  # second call: from the parameter cache
        assert torch.equal(a, out0) and torch.equal(b, out0), name
    assert torch.equal(hydragen_attention_nopad(q, k, v, [sk], [sv], seq_len=lens), out0)


@pytest.mark.parametrize("graph", [False, True])
def test_model_schedules_ragged_unique_prompts_longest_first(graph):
    """generate() with per-sequence unique prompts of different lengths: the decode loop runs under the longest-first schedule
    (and, graphed, keeps replaying ONE captured graph while the schedule's contents change between generations); logits and tokens
    are those of the unscheduled run, bit for bit."""
    from tests.test_model_gpu import make_model

    model = make_model(torch.bfloat16, head_dim=128, kv_heads=4)
    model.graph(graph)
    g = torch.Generator(device=DEV).manual_seed(9)
    rnd = lambda *s: torch.randint(1, 512, s, device=DEV, generator=g)
    ids = [rnd(1, 20), rnd(6, 12)]
    new = 5
    model.setup_caches(max_unique_batch_size=6, max_unique_seq_length=32, max_shared_batch_sizes=[1], max_shared_seq_lengths=[20])
    overrides = rnd(6, new)
    runs = {}
    for lens_u in ([12, 3, 9, 1, 12, 5], [2, 11, 4, 12, 7, 7]):
        lens = [torch.tensor([20], device=DEV), torch.tensor(lens_u, device=DEV)]
        for on in (True, False, True):
            model.schedule_longest_first = on
            out, logits = model.generate(input_ids=ids, seq_lens=lens, num_return_sequences=1, max_new_tokens=new, temperature=0.0,
                                         return_logits=True, token_overrides=overrides)
            if on:
                want = torch.argsort(torch.tensor(lens_u), descending=True, stable=True).to(torch.int32)
                assert torch.equal(model.seq_order_buf[:6].cpu(), want)
            key = tuple(lens_u)
            if key in runs:
                assert torch.equal(out, runs[key][0]) and all(torch.equal(x, y) for x, y in zip(logits, runs[key][1])), (lens_u, on)
            runs[key] = (out, logits)
    model.schedule_longest_first = True
