import sys; sys.path.insert(0, "/root/repo")
import torch
from tests.test_model_gpu import make_model, DEV
from hydragen_amd import layer_ops
for dtype in (torch.float16, torch.bfloat16):
    for seed in range(6):
        torch.manual_seed(seed)
        model = make_model(dtype, layers=3, seed=seed)
        m = model.model
        ids = torch.randint(1, model.config.vocab_size, (5, 9), device=DEV)
        pos = torch.arange(9, device=DEV)[None].expand(5, 9).contiguous()
        kw = dict(max_unique_batch_size=5, max_unique_seq_length=32, max_shared_batch_sizes=[], max_shared_seq_lengths=[])
        model.setup_caches(**kw); model.set_mode("unique-prefill")
        with torch.no_grad():
            fused = m(ids, pos)
            sup = layer_ops.supported; layer_ops.supported = lambda *a, **k: False
            model.setup_caches(**kw); plain = m(ids, pos); layer_ops.supported = sup
        d = (fused.float() - plain.float())
        print(dtype, seed, "max", d.abs().max().item(), "scale", plain.float().abs().max().item(), "relL2", (d.norm() / plain.float().norm()).item())
