"""-m gpu: seeded random hierarchies through the whole operator against the float64 oracle.
Each case draws: number of shared levels (0-3), per-level group counts that divide the batch, uniform or ragged
(varlen, packed) prefix lengths from 1 to ~700 keys (crossing the 32-key block and 128-key tile boundaries of the
prefix kernel and its split-KV plan), ragged unique lengths incl. 1 and the full cache, GQA ratio, head dim,
dtype.  Decode cases (nq = 1, seq_lens given) go through the fused entry point; a quarter of the cases are
prefill-shaped (nq > 1, seq_lens = None: causal unique pass, attention.py:343-345)."""
import numpy as np
import pytest
import torch

from oracle import hydragen_oracle as O
from tests.cases import make_case
from tests.gpu_util import assert_close_l2, case_to_device

pytestmark = pytest.mark.gpu


def _draw(seed):
    rng = np.random.default_rng(10_000 + seed)
    B = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24]))
    nlev = int(rng.integers(0, 4))
    divs = [d for d in (1, 2, 3, 4, 6, 8, 12) if B % d == 0]
    sbs = sorted(int(rng.choice(divs)) for _ in range(nlev))
    sizes = []
    for sb in sbs:
        top = int(rng.choice([40, 140, 300, 700]))
        if rng.random() < 0.5:
            sizes.append([int(rng.integers(1, top))] * sb)
        else:
            sizes.append([int(rng.integers(1, top)) for _ in range(sb)])
    prefill = seed % 4 == 3
    if prefill:
        nq = int(rng.integers(2, 7))
        uniq = [nq] * B  # causal unique pass: the unique keys are the nq new tokens
    else:
        nq = 1
        cap = int(rng.choice([1, 5, 17, 40, 130]))
        uniq = [int(rng.integers(1, cap + 1)) for _ in range(B)]
        uniq[int(rng.integers(0, B))] = cap
    kvh = int(rng.choice([1, 2, 4]))
    g = int(rng.choice([1, 2, 4, 8]))
    dim = int(rng.choice([64, 128]))
    dt = "bf16" if rng.random() < 0.5 else "f16"
    return dict(sizes=sizes + [uniq], qheads=kvh * g, kvheads=kvh, dim=dim, dtype=dt, seed=500 + seed, nq=nq,
                force_seq_lens=not prefill), prefill


check_fuzz = assert_close_l2


@pytest.mark.parametrize("seed", list(range(48)) + [1000 + i for i in range(16)])
def test_random_hierarchy_vs_oracle(seed):
    from hydragen_amd.attention import hydragen_attention

    kw, prefill = _draw(seed % 1000)
    if seed >= 1000:  # the same hierarchies at head_dim 256 (one query block per wave / 32 lanes per key)
        kw["dim"] = 256
    case = make_case(**kw)
    if prefill:
        case["seq_lens"] = None
    if not case["shared_ks"] and case["k"].shape[1] == 0:
        pytest.skip("empty problem")
    d = case_to_device(case)
    out = hydragen_attention(**d)
    torch.cuda.synchronize()
    want = O.hydragen_attention(case["q"], case["k"], case["v"], case["shared_ks"], case["shared_vs"],
                                case["shared_cu_seq_lens"], case["shared_max_seq_lens"], case["use_varlens"],
                                case["seq_lens"])
    check_fuzz(out.float().cpu().numpy(), want, case["dtype"], f"seed {seed}: {kw['sizes']} {kw}")


@pytest.mark.parametrize("seed", range(40))
def test_random_primitives_vs_oracle(seed):
    """The three primitives on their own with random shapes: dense / causal `flash_attention` (key counts from 1 to
    ~1500: split-KV plans, ragged last blocks, bottom-right causal alignment with sq != sk), packed
    `flash_attention_varlen`, and `flash_attention_seqlen` (both suffix kernels, empty sequences included); out AND
    LSE against the float64 oracle."""
    from hydragen_amd.flash import flash_attention, flash_attention_seqlen, flash_attention_varlen
    from tests.gpu_util import dev
    from tests.cases import _round

    rng = np.random.default_rng(77_000 + seed)
    dt = "bf16" if seed & 1 else "f16"
    D = int(rng.choice([64, 128]))
    hkv = int(rng.choice([1, 2, 4]))
    hq = hkv * int(rng.choice([1, 2, 4, 8]))
    rnd = lambda *s: _round(rng.standard_normal(s, dtype=np.float32), dt)
    lse_ok = lambda got, want: (np.abs(got - want) <= 2e-3 + 1e-5 * np.abs(want)).all()
    kind = seed % 3
    if kind == 0:
        b, sk = int(rng.integers(1, 4)), int(rng.choice([1, 17, 100, 129, 400, 1500]))
        causal = bool(rng.integers(0, 2))
        sq = int(rng.integers(1, min(sk, 200) + 1)) if causal else int(rng.integers(1, 300))
        q, k, v = rnd(b, sq, hq, D), rnd(b, sk, hkv, D), rnd(b, sk, hkv, D)
        out, lse = flash_attention(dev(q, dt), dev(k, dt), dev(v, dt), causal=causal)
        torch.cuda.synchronize()
        want, wlse = O.flash_attention(q, k, v, causal=causal)
        check_fuzz(out.float().cpu().numpy(), want, dt, f"flash_attention seed {seed} b={b} sq={sq} sk={sk} causal={causal}")
        assert lse_ok(lse.cpu().numpy(), wlse)
    elif kind == 1:
        n = int(rng.integers(1, 6))
        qlens = [int(rng.integers(1, 150)) for _ in range(n)]
        klens = [int(rng.integers(1, 500)) for _ in range(n)]
        cu_q = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
        cu_k = np.concatenate([[0], np.cumsum(klens)]).astype(np.int32)
        q, k, v = rnd(cu_q[-1], hq, D), rnd(cu_k[-1], hkv, D), rnd(cu_k[-1], hkv, D)
        out, lse = flash_attention_varlen(dev(q, dt), dev(k, dt), dev(v, dt), dev(cu_q), dev(cu_k), max(qlens), max(klens))
        torch.cuda.synchronize()
        want, wlse = O.flash_attention_varlen(q, k, v, cu_q, cu_k, max(qlens), max(klens))
        check_fuzz(out.float().cpu().numpy(), want, dt, f"varlen seed {seed} q={qlens} k={klens}")
        got_lse = lse.cpu().numpy()
        for i, ql in enumerate(qlens):  # only the valid part is defined (attention.py:333-338)
            assert lse_ok(got_lse[i, :, :ql], wlse[i, :, :ql])
    else:
        b, nq, mk = int(rng.integers(1, 40)), int(rng.choice([1, 1, 2, 3])), int(rng.choice([1, 9, 40, 130, 300]))
        lens = rng.integers(0, mk + 1, b).astype(np.int32)
        lens[int(rng.integers(0, b))] = mk
        q, k, v = rnd(b, nq, hq, D), rnd(b, mk, hkv, D), rnd(b, mk, hkv, D)
        out, lse = flash_attention_seqlen(dev(q, dt), dev(k, dt), dev(v, dt), seq_len=dev(lens))
        torch.cuda.synchronize()
        want, wlse = O.flash_attention_seqlen(q, k, v, lens)
        nz = lens > 0  # attention over zero keys is undefined; the kernel returns 0 / -inf there
        check_fuzz(out.float().cpu().numpy()[nz], want[nz], dt, f"seqlen seed {seed} b={b} nq={nq} mk={mk} {hq}/{hkv}")
        assert lse_ok(lse.cpu().numpy()[nz], wlse[nz])
