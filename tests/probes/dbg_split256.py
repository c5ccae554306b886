import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import torch
from hydragen_amd import _lib
from hydragen_amd._lib import PrefixParams, HYD_LSE_BQH
from hydragen_amd.flash import _dtype_code
lib = _lib.load()
torch.manual_seed(0)
for D, L1 in ((256, 199), (256, 384), (256, 385), (256, 10)):
    B, Hq, Hkv = 6, 8, 1
    q = torch.randn(B, 1, Hq, D, device="cuda", dtype=torch.float16)
    L = [652, L1]
    k = torch.randn(sum(L), Hkv, D, device="cuda", dtype=torch.float16); v = torch.randn_like(k)
    cu = torch.tensor([0, 652, 652 + L1], device="cuda", dtype=torch.int32)
    out = torch.full_like(q, 7.0); lse = torch.full((B, 1, Hq), 7.0, device="cuda")
    p = PrefixParams()
    p.q, p.k, p.v, p.out, p.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr()
    p.cu_seqlens_k = cu.data_ptr()
    p.k_tok_stride, p.k_head_stride, p.v_tok_stride, p.v_head_stride = k.stride(0), k.stride(1), v.stride(0), v.stride(1)
    p.dtype = _dtype_code(q); p.B, p.nq, p.Hq, p.Hkv, p.D = B, 1, Hq, Hkv, D
    p.sb, p.kv_len, p.lse_layout, p.num_splits = 2, 652, HYD_LSE_BQH, 2
    ns, grid, sl = C.c_int32(), C.c_int32(), C.c_int32()
    lib.hyd_prefix_plan(C.byref(p), C.byref(ns), C.byref(grid), C.byref(sl))
    n = lib.hyd_prefix_workspace_bytes(C.byref(p))
    ws = torch.full((n // 4,), 12345.0, device="cuda", dtype=torch.float32)
    p.workspace, p.workspace_bytes = ws.data_ptr(), n
    _lib.check(lib.hyd_prefix_attn_fwd(C.byref(p), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    rows = B * Hq
    ob = (rows * D * 4 + 255) // 256 * 256 // 4
    lb = (rows * 4 + 255) // 256 * 256 // 4
    print(f"D={D} L1={L1} plan: splits {ns.value} grid {grid.value} split_len {sl.value}; out finite {bool(torch.isfinite(out).all())}")
    for s in range(ns.value):
        o = ws[s * ob: s * ob + rows * D].view(B, Hq, D)
        l = ws[ns.value * ob + s * lb: ns.value * ob + s * lb + rows].view(B, Hq)
        for b in (0, 3):
            print(f"  slice {s} seq {b}: untouched {int((o[b] == 12345.0).sum())} nan {int(torch.isnan(o[b]).sum())} absmax {float(o[b][torch.isfinite(o[b])].abs().max()) if torch.isfinite(o[b]).any() else -1:.3g} lse {l[b, :3].tolist()} first {o[b, 0, :4].tolist()}")
