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
from tests.gpu_util import ATOL, case_to_device

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


REL_L2 = {"f16": 1e-3, "bf16": 8e-3}


def check_fuzz(got, want, dtype, what):
    """Random small hierarchies are judged by the maximum absolute error (the reference's atol, x 2^3 for bf16) and by
    the relative L2 error ||got - want|| / ||want|| <= 1e-3 for fp16 (the figure BASELINE.json states), x 2^3 for bf16.
    The reference's third figure, the MEAN of the element-wise relative difference (tests/test_attention.py:183-185),
    is kept for the reference-shaped cases of test_parity_gpu.py but is not a stable statistic here: outputs near zero
    dominate it on tensors of a few hundred elements (a 3000-seed soak, tools/soak.py: relative L2 error peaks at
    3.8e-4 / 3.3e-3, while the mean relative difference of the same runs scatters up to 1.6e-2 for fp16 cases whose
    largest absolute error is 5e-5)."""
    got = np.asarray(got, dtype=np.float64)
    assert got.shape == want.shape and np.isfinite(got).all(), what
    err = np.abs(got - want).max()
    l2 = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
    assert err <= ATOL[dtype] and l2 <= REL_L2[dtype], f"{what}: max abs {err:.3e} relative L2 {l2:.3e}"
    return err, l2


@pytest.mark.parametrize("seed", range(48))
def test_random_hierarchy_vs_oracle(seed):
    from hydragen_amd.attention import hydragen_attention

    kw, prefill = _draw(seed)
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
