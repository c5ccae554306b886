"""Decoder-layer glue kernels (hyd_add_rmsnorm, hyd_swiglu) through the C ABI against fp32 torch formulas of
transformers' LlamaRMSNorm / LlamaMLP as the reference uses them (/root/reference/hydragen/llama.py:2-6,604-631)."""
import pytest
import torch

from hydragen_amd import layer_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ulp(dtype):
    return 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,n", [(1024, 4096), (3, 8192), (1, 8), (7, 520), (64, 16384), (5, 5120), (0, 64)])
@pytest.mark.parametrize("with_residual", [True, False])
def test_add_rmsnorm_vs_fp32(dtype, rows, n, with_residual):
    g = torch.Generator(device=DEV).manual_seed(rows * 131 + n)
    x = torch.randn(rows, n, device=DEV, generator=g).to(dtype)
    r = (3 * torch.randn(rows, n, device=DEV, generator=g)).to(dtype) if with_residual else None
    w = (1 + 0.1 * torch.randn(n, device=DEV, generator=g)).to(dtype)
    eps = 1e-5
    summed, normed = layer_ops.add_rms_norm(x, r, w, eps)
    if with_residual:
        want_sum = (x.float() + r.float()).to(dtype)  # the stored residual stream: one rounding
        assert torch.equal(summed, want_sum)
    else:
        want_sum = x
        assert summed is x
    h = want_sum.float()
    want = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps) * w.float()
    assert normed.shape == x.shape and normed.dtype == dtype
    if rows:
        err = (normed.float() - want).abs()
        assert (err <= _ulp(dtype) * want.abs() + 1e-6).all(), float((err / (want.abs() + 1e-6)).max())
        # transformers' LlamaRMSNorm rounds the normalised value to the 16-bit dtype before the weight multiply (two
        # roundings); the kernel rounds once.  Stated deviation: the two agree to two 16-bit ulps everywhere, and the
        # kernel is never farther from the fp32 formula than the two-rounding form is.
        hf = w * (h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)).to(dtype)
        assert ((normed.float() - hf.float()).abs() <= 2 * _ulp(dtype) * want.abs() + 1e-6).all()
        assert float(err.mean()) <= float((hf.float() - want).abs().mean()) + 1e-9


def test_add_rmsnorm_views_and_3d():
    dtype = torch.bfloat16
    big = torch.randn(16, 2, 3 * 256, device=DEV).to(dtype)
    x = big[..., 256:512]  # a column slice: row stride 768
    r = torch.randn(16, 2, 256, device=DEV).to(dtype)
    w = torch.ones(256, device=DEV, dtype=dtype)
    s, nrm = layer_ops.add_rms_norm(x, r, w, 1e-6)
    h = (x.float() + r.float()).to(dtype)
    assert torch.equal(s, h) and s.shape == (16, 2, 256)
    want = torch.nn.functional.rms_norm(h.float(), (256,), None, 1e-6)
    assert (nrm.float() - want).abs().max() <= 2.0 ** -8 * want.abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,n", [(1024, 11008), (2, 28672), (1, 8), (33, 200), (0, 16)])
def test_swiglu_vs_fp32(dtype, rows, n):
    g = torch.Generator(device=DEV).manual_seed(n)
    gu = (2 * torch.randn(rows, 2 * n, device=DEV, generator=g)).to(dtype)
    gate, up = gu.split(n, dim=-1)  # the fused GEMM output's column halves
    out = layer_ops.swiglu(gate, up)
    want = torch.nn.functional.silu(gate.float()) * up.float()
    assert out.shape == (rows, n) and out.dtype == dtype and out.is_contiguous()
    if rows:
        err = (out.float() - want).abs()
        assert (err <= _ulp(dtype) * want.abs() + 1e-6).all(), float(err.max())
    # extremes: silu(-inf-ish) = 0, silu(large) = x
    e = torch.tensor([[-60000.0, 60000.0, 0.0, -0.0, 1e-4, -20.0, 20.0, 3.0]], device=DEV).to(dtype)
    o = layer_ops.swiglu(e, torch.ones_like(e))
    assert torch.isfinite(o).all()
    assert (o.float() - torch.nn.functional.silu(e.float())).abs().max() <= 60000 * _ulp(dtype)


def test_refuses_what_it_cannot_take():
    x = torch.randn(4, 12, device=DEV, dtype=torch.bfloat16)  # 12 is not a multiple of 8
    assert not layer_ops.supported(x)
    with pytest.raises(AssertionError):
        layer_ops.add_rms_norm(x, None, torch.ones(12, device=DEV, dtype=torch.bfloat16), 1e-6)
    with pytest.raises(RuntimeError):
        layer_ops.swiglu(torch.zeros(1, 8, dtype=torch.bfloat16), torch.zeros(1, 8, dtype=torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_sample_tokens_temperature_zero_is_argmax(dtype):
    g = torch.Generator(device=DEV).manual_seed(1)
    for rows, n in [(64, 32000), (3, 50), (1, 1), (5, 4099)]:
        logits = torch.randn(rows, n, device=DEV, generator=g).to(dtype)
        got = layer_ops.sample_tokens(logits, 0.0)
        assert got.shape == (rows, 1) and got.dtype == torch.int64
        assert torch.equal(got[:, 0], logits.float().argmax(-1))
    tie = torch.zeros(2, 300, device=DEV, dtype=dtype)
    tie[0, 17] = tie[0, 200] = 1.0  # ties: the lowest index
    assert layer_ops.sample_tokens(tie, 0.0)[:, 0].tolist() == [17, 0]
    view = torch.randn(4, 2, 1000, device=DEV).to(dtype)[:, -1]  # a strided [B, V] view (the model's last position)
    assert torch.equal(layer_ops.sample_tokens(view, 0.0)[:, 0], view.float().argmax(-1))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_sample_tokens_draws_from_the_softmax(dtype):
    """Chi-square of 65536 draws against softmax(logits / T) on a 20-token vocabulary (19 degrees of freedom: the
    0.999 quantile is 43.8), masked (-inf) tokens never drawn, determinism for a key, generator semantics."""
    n, rows, T = 20, 65536, 1.7
    row = torch.linspace(-3, 3, n, device=DEV)
    row[5] = -float("inf")
    logits = row.to(dtype)[None].expand(rows, n).contiguous()
    p = torch.softmax(logits[0].float() / T, -1)
    tok = layer_ops.sample_tokens(logits, T, key=(1234, 8))[:, 0]
    counts = torch.bincount(tok, minlength=n).float()
    assert counts[5] == 0
    keep = p > 0
    chi2 = (((counts - rows * p) ** 2)[keep] / (rows * p[keep])).sum().item()
    assert chi2 < 43.8, chi2
    # rows are independent draws: neighbouring rows agree no more often than chance
    agree = (tok[1:] == tok[:-1]).float().mean().item()
    assert abs(agree - float((p * p).sum())) < 0.01
    assert torch.equal(tok, layer_ops.sample_tokens(logits, T, key=(1234, 8))[:, 0])
    assert not torch.equal(tok, layer_ops.sample_tokens(logits, T, key=(1234, 12))[:, 0])
    assert not torch.equal(tok, layer_ops.sample_tokens(logits, T, key=(1235, 8))[:, 0])
    torch.manual_seed(7)
    a1, a2 = layer_ops.sample_tokens(logits, T), layer_ops.sample_tokens(logits, T)
    torch.manual_seed(7)
    b1 = layer_ops.sample_tokens(logits, T)
    assert torch.equal(a1, b1) and not torch.equal(a1, a2)


def test_sample_tokens_high_temperature_is_uniform():
    logits = torch.randn(32768, 16, device=DEV).bfloat16()
    counts = torch.bincount(layer_ops.sample_tokens(logits, 1e4, key=(3, 0))[:, 0], minlength=16).float()
    chi2 = (((counts - 2048) ** 2) / 2048).sum().item()
    assert chi2 < 37.7, chi2  # 15 degrees of freedom, 0.999 quantile
