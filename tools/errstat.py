import sys, torch
sys.path.insert(0, "/root/repo")
from hydragen_amd.flash import flash_attention
torch.manual_seed(0)
B, P, H, D = 256, 2048, 8, 128
for dt in (torch.bfloat16, torch.float16):
    q = torch.randn(1, B, H, D, device="cuda", dtype=dt)
    k = torch.randn(1, P, H, D, device="cuda", dtype=dt)
    v = torch.randn(1, P, H, D, device="cuda", dtype=dt)
    out, lse = flash_attention(q, k, v)
    qf, kf, vf = q.double(), k.double(), v.double()
    s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * D ** -0.5
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vf)
    reflse = torch.logsumexp(s, -1)  # b h q
    e = (out.double() - ref).abs()
    rd = (2 * e / (out.double().abs() + ref.abs() + 1e-8)).mean().item()
    print(dt, "max abs", e.max().item(), "mean abs", e.mean().item(), "mean rdiff", rd, "lse max err", (lse.double() - reflse).abs().max().item())
    # spiky keys: force rescale events late in the sequence
    k2 = k.clone(); k2[:, 1500] *= 6; k2[:, 1900] *= 9
    out2, lse2 = flash_attention(q, k2, v)
    s = torch.einsum("bqhd,bkhd->bhqk", qf, k2.double()) * D ** -0.5
    ref2 = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vf)
    e = (out2.double() - ref2).abs()
    print("   spiky: max abs", e.max().item(), "mean abs", e.mean().item(), "lse max err", (lse2.double() - torch.logsumexp(s, -1)).abs().max().item())
