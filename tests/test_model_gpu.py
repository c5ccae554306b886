"""-m gpu: the model shell (hydragen_amd/llama.py) around the HIP attention path, with random weights.
Mirrors the reference's integration tests (tests/test_e2e.py): per-step decode logits against a plain
causal transformer evaluated on the concatenated prompt (there: HF transformers; here: a torch fp32
re-evaluation with the same weights), with token_overrides forcing identical tokens
(tests/test_e2e.py:104-119, bounds :29-30), plus the A/B invariants hydragen vs no-sharing
(:122-210) and hierarchy vs flattened (:213-298), and HIP-graph on/off."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rdiff(a, b, eps=1e-8):
    return 2 * (a - b).abs() / (a.abs() + b.abs() + eps)


def make_model(dtype, head_dim=64, kv_heads=2, layers=2, seed=0):
    from hydragen_amd.llama import HydragenLlamaForCausalLM, LlamaConfig

    cfg = LlamaConfig(hidden_size=4 * head_dim, intermediate_size=512, num_hidden_layers=layers,
                      num_attention_heads=4, num_key_value_heads=kv_heads, vocab_size=512,
                      max_position_embeddings=1024, rms_norm_eps=1e-5)
    return HydragenLlamaForCausalLM.from_config(cfg, dtype=dtype, device=DEV, seed=seed, std=0.05)


@torch.no_grad()
def ref_logits(model, ids):
    """fp32 causal transformer over full sequences ids [b, n] with the model's weights."""
    from hydragen_amd.llama import rotate_half

    cfg = model.config
    H, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    D = cfg.hidden_size // H
    f = lambda t: t.float()
    b, n = ids.shape
    h = f(model.model.embed_tokens.weight)[ids]
    pos = torch.arange(n, device=ids.device)
    cos = model.model.rotary_emb.cos_cached[pos][None, :, None, :]
    sin = model.model.rotary_emb.sin_cached[pos][None, :, None, :]

    def norm(x, w, eps):
        return f(w) * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))

    for layer in model.model.layers:
        a = layer.self_attn
        x = norm(h, layer.input_layernorm.weight, cfg.rms_norm_eps)
        q = (x @ f(a.q_proj.weight).T).view(b, n, H, D)
        k = (x @ f(a.k_proj.weight).T).view(b, n, Hkv, D)
        v = (x @ f(a.v_proj.weight).T).view(b, n, Hkv, D)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        k = k.repeat_interleave(H // Hkv, 2)
        v = v.repeat_interleave(H // Hkv, 2)
        s = torch.einsum("bqhd,bkhd->bhqk", q, k) * D ** -0.5
        s = s.masked_fill(torch.triu(torch.ones(n, n, device=ids.device, dtype=torch.bool), 1), float("-inf"))
        o = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v).reshape(b, n, H * D)
        h = h + o @ f(a.o_proj.weight).T
        x = norm(h, layer.post_attention_layernorm.weight, cfg.rms_norm_eps)
        m = layer.mlp
        h = h + (torch.nn.functional.silu(x @ f(m.gate_proj.weight).T) * (x @ f(m.up_proj.weight).T)) @ f(m.down_proj.weight).T
    h = norm(h, model.model.norm.weight, cfg.rms_norm_eps)
    return h @ f(model.lm_head.weight).T


def test_rope_append_kernel_vs_torch():
    from hydragen_amd.fused_decode import rope_append_decode
    from hydragen_amd.llama import RotaryTable, apply_rotary_pos_emb

    for dtype, D, Hq, Hkv in ((torch.bfloat16, 128, 8, 2), (torch.float16, 64, 4, 4)):
        B, maxS = 37, 48
        g = torch.Generator(device=DEV).manual_seed(1)
        qkv = torch.randn(B, 1, (Hq + 2 * Hkv) * D, device=DEV, dtype=dtype, generator=g)
        q = qkv[..., : Hq * D].view(B, 1, Hq, D)                    # views of one projection output:
        k = qkv[..., Hq * D : (Hq + Hkv) * D].view(B, 1, Hkv, D)    # non-trivial batch strides
        v = qkv[..., (Hq + Hkv) * D :].view(B, 1, Hkv, D)
        rot = RotaryTable(D, 512, 10000.0, device=DEV)
        shared = torch.randint(0, 100, (B,), device=DEV, generator=g)
        idx = torch.randint(0, maxS, (B,), device=DEV, generator=g)
        pos = (shared + idx)[:, None]
        kc = torch.zeros(B + 3, maxS, Hkv, D, device=DEV, dtype=dtype)
        vc = torch.zeros_like(kc)
        qo, sl = rope_append_decode(q, k, v, rot.cos_cached, rot.sin_cached, pos, shared, kc, vc)
        qr, kr = apply_rotary_pos_emb(q.float(), k.float(), rot.cos_cached, rot.sin_cached, pos)
        assert torch.equal(sl.long(), idx + 1)
        tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
        assert (qo.float() - qr).abs().max() < tol
        bi = torch.arange(B, device=DEV)
        assert (kc[bi, idx].float() - kr[:, 0]).abs().max() < tol
        assert torch.equal(vc[bi, idx], v[:, 0])
        mask = torch.ones_like(kc, dtype=torch.bool)
        mask[bi, idx] = False
        assert kc[mask].abs().max() == 0 and vc[mask].abs().max() == 0   # nothing else was touched


def test_decode_logits_head_dim_256():
    """A model with 256-wide heads (flash-attn's upper limit, flash.py:295-304): the D = 256 instantiations of the prefix
    pass, the suffix pass and the fused RoPE + append kernel under the decode graph."""
    test_decode_logits_vs_fp32_transformer(torch.bfloat16, True, "three-level", head_dim=256)


@pytest.mark.parametrize("spec", ["prefix+completions", "three-level", "padded-shared"])
def test_decode_graph_with_the_two_stream_form_in_every_layer(spec):
    """The decode graph captured with the two-stream form forced on (a fork and a join inside every layer's attention,
    shared phase on a side stream) and with fp32 prefix partials: same logits check as the one-call form."""
    from hydragen_amd import attention as A

    prev, prev_f32 = A.set_two_stream("on"), A.set_f32_partials(True)
    try:
        test_decode_logits_vs_fp32_transformer(torch.bfloat16, True, spec)
    finally:
        A.set_two_stream(prev)
        A.set_f32_partials(prev_f32)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("spec", ["prefix+completions", "three-level", "padded-shared", "prefix+suffix"])
def test_decode_logits_vs_fp32_transformer(dtype, graph, spec, head_dim=None):
    model = make_model(dtype, head_dim=head_dim or (64 if dtype == torch.float16 else 128))
    model.graph(graph)
    g = torch.Generator(device=DEV).manual_seed(3)
    rnd = lambda *s: torch.randint(1, 512, s, device=DEV, generator=g)
    new = 6
    if spec == "prefix+completions":
        ids, lens, nret = [rnd(1, 40)], None, 4
    elif spec == "three-level":
        ids, lens, nret = [rnd(1, 24), rnd(2, 9), rnd(4, 5)], None, 2
    elif spec == "padded-shared":
        ids, nret = [rnd(1, 30), rnd(2, 12)], 3
        lens = [torch.tensor([30], device=DEV), torch.tensor([12, 7], device=DEV)]
    else:
        ids, lens, nret = [rnd(1, 33), rnd(4, 10)], None, 1
    B = ids[-1].shape[0] * nret
    model.setup_caches(max_unique_batch_size=B, max_unique_seq_length=64,
                       max_shared_batch_sizes=[x.shape[0] for x in ids], max_shared_seq_lengths=[x.shape[1] for x in ids])
    overrides = rnd(B, new)
    out, logits = model.generate(input_ids=ids, seq_lens=lens, num_return_sequences=nret, max_new_tokens=new,
                                 temperature=0.0, return_logits=True, token_overrides=overrides)
    assert out.shape == (B, new) and len(logits) == new
    # reference: every completion's full token sequence through a plain causal transformer
    for j in range(B):
        parts = []
        for li, x in enumerate(ids):
            row = j // (B // x.shape[0])
            n = x.shape[1] if lens is None else int(lens[li][row])
            parts.append(x[row, :n])
        parts.append(overrides[j, : new - 1])
        full = torch.cat(parts)[None]
        ref = ref_logits(model, full)[0]
        nprompt = full.shape[1] - (new - 1)
        got = torch.stack([l[j] for l in logits])                     # logits after prompt, tok1, ...
        want = ref[nprompt - 1 :]
        # fp16: the reference's bounds (tests/test_e2e.py:29-30,117-119).  bf16 (3 fewer mantissa bits, small
        # random-weight logits): same absolute bound, mean relative bound scaled accordingly.
        assert (got - want).abs().max() < 0.75, spec
        assert rdiff(got, want).mean() < (0.05 if dtype == torch.float16 else 0.15), spec


def test_generate_room_check_is_per_sequence():
    """Right-padded shared prompts of lengths [9, 5], fan-out, 16 new tokens, a unique cache of exactly 16 tokens: every
    sequence's last cache index is 14, so the call fits (the reference, llama.py:1156-1396, has no such check at all).
    A check built from max(position) - min(shared length) refused it."""
    model = make_model(torch.bfloat16, head_dim=128)
    g = torch.Generator(device=DEV).manual_seed(11)
    ids = [torch.randint(1, 512, (2, 9), device=DEV, generator=g)]
    lens = [torch.tensor([9, 5], device=DEV)]
    model.setup_caches(max_unique_batch_size=4, max_unique_seq_length=16, max_shared_batch_sizes=[2], max_shared_seq_lengths=[9])
    out = model.generate(input_ids=ids, seq_lens=lens, num_return_sequences=2, max_new_tokens=16, temperature=0.0)
    assert out.shape == (4, 16)
    with pytest.raises(ValueError, match="unique cache holds"):
        model.generate(input_ids=ids, seq_lens=lens, num_return_sequences=2, max_new_tokens=18, temperature=0.0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_hydragen_vs_nosharing_and_flat_hierarchy(dtype):
    # fp16: the reference's bound (tests/test_e2e.py:210,298); bf16: scaled for its 3 fewer mantissa bits
    bound = 0.02 if dtype == torch.float16 else 0.08
    model = make_model(dtype, head_dim=128, kv_heads=4)
    g = torch.Generator(device=DEV).manual_seed(5)
    rnd = lambda *s: torch.randint(1, 512, s, device=DEV, generator=g)
    prefix, new, nret = rnd(1, 50), 8, 6
    overrides = rnd(nret, new)
    model.setup_caches(max_unique_batch_size=nret, max_unique_seq_length=64 + 16, max_shared_batch_sizes=[1],
                       max_shared_seq_lengths=[50])
    kw = dict(input_ids=prefix, num_return_sequences=nret, max_new_tokens=new, temperature=0.0, return_logits=True,
              token_overrides=overrides)
    _, a = model.generate(**kw)
    _, b = model.generate(disable_hydragen=True, **kw)                 # tests/test_e2e.py:122-210
    assert rdiff(torch.stack(a), torch.stack(b)).mean() < bound
    # hierarchy vs flattened: prefix + 2 second-level prompts + completions   (tests/test_e2e.py:213-298)
    mid = rnd(2, 11)
    overrides = rnd(2 * nret, new)
    model.setup_caches(max_unique_batch_size=2 * nret, max_unique_seq_length=64, max_shared_batch_sizes=[1, 2],
                       max_shared_seq_lengths=[50, 11])
    kw = dict(input_ids=[prefix, mid], num_return_sequences=nret, max_new_tokens=new, temperature=0.0,
              return_logits=True, token_overrides=overrides)
    _, a = model.generate(**kw)
    _, b = model.generate(disable_hierarchy=True, **kw)
    assert rdiff(torch.stack(a), torch.stack(b)).mean() < bound
    # fused RoPE/append kernel vs the torch scatter path
    for layer in model.model.layers:
        layer.self_attn.use_fused_decode = False
    _, c = model.generate(**kw)
    assert rdiff(torch.stack(a), torch.stack(c)).mean() < bound


def _tp_gpu_worker(rank, world, port, shard_dir, ids_cpu, overrides_cpu, ret, xgmi=False):
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    import torch.distributed as dist
    from hydragen_amd import tp, utils

    # both ranks share the one GPU of the test box; the collective goes through gloo (RCCL needs one GPU per rank)
    assert utils.maybe_init_dist(backend="gloo") == rank
    model = tp.from_pretrained_tp(shard_dir, device=DEV)
    comm = None
    if xgmi:
        # the direct all-reduce over IPC-mapped peer blocks (hyd_allreduce_sum), captured INSIDE the decode graph like
        # the reference's NCCL all-reduce (llama.py:849-854); gloo only carries the handle exchange
        from hydragen_amd.xgmi_allreduce import XgmiAllReduce
        comm = XgmiAllReduce(max_bytes=1 << 20)
        tp.use_xgmi_allreduce(comm)
        model.graph(True)
    ids = [x.to(DEV) for x in ids_cpu]
    B, new = overrides_cpu.shape
    model.setup_caches(max_unique_batch_size=B, max_unique_seq_length=32,
                       max_shared_batch_sizes=[x.shape[0] for x in ids], max_shared_seq_lengths=[x.shape[1] for x in ids])
    out, logits = model.generate(input_ids=ids, seq_lens=None, num_return_sequences=B // ids[-1].shape[0], max_new_tokens=new,
                                 temperature=0.0, return_logits=True, token_overrides=overrides_cpu.to(DEV))
    if rank == 0:
        ret.put(torch.stack(logits).float().cpu().numpy())
    if comm is not None:
        assert comm.status() == 0
    dist.barrier()
    if comm is not None:
        tp.use_xgmi_allreduce(None)
        comm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("xgmi", [False, True])
def test_tensor_parallel_model_two_ranks_on_one_gpu(tmp_path, xgmi):
    """head-sharded model shell (apply_tp shards, per-rank KV caches with Hkv/2 heads, the HIP
    operator on each rank's heads, all-reduce after o_proj and down_proj) reproduces the unsharded model's decode
    logits.  World size 2 on the single GPU of the box, gloo collectives."""
    import socket
    import torch.multiprocessing as mp
    from hydragen_amd import tp

    dtype = torch.float16
    model = make_model(dtype, head_dim=64)
    tp.make_tp_files(model, tmp_path, num_splits=2)
    g = torch.Generator(device=DEV).manual_seed(11)
    rnd = lambda *s: torch.randint(1, 512, s, device=DEV, generator=g)
    ids, nret, new = [rnd(1, 37), rnd(2, 9)], 3, 5
    B = ids[-1].shape[0] * nret
    overrides = rnd(B, new)
    model.setup_caches(max_unique_batch_size=B, max_unique_seq_length=32,
                       max_shared_batch_sizes=[x.shape[0] for x in ids], max_shared_seq_lengths=[x.shape[1] for x in ids])
    _, logits = model.generate(input_ids=ids, seq_lens=None, num_return_sequences=nret, max_new_tokens=new,
                               temperature=0.0, return_logits=True, token_overrides=overrides)
    want = torch.stack(logits).float().cpu().numpy()

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_tp_gpu_worker, args=(r, 2, port, str(tmp_path), [x.cpu() for x in ids], overrides.cpu(), ret, xgmi))
             for r in range(2)]
    for p in procs:
        p.start()
    got = ret.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # two different summation orders of fp16 partial products (per-rank o_proj / down_proj halves, then the sum)
    assert abs(got - want).max() < 0.05


@pytest.mark.parametrize("graph", [False, True])
def test_readme_cache_workflows(graph):
    """The finer-grained control flows of the reference README (README.md:183-289): (a) shared_cache_op="extend" on
    the first generate, later calls reuse the cached prompt through `starting_logits`; (b) manual `append_shared` +
    `starting_logits`; both must give the logits of a plain causal transformer over prompt + forced tokens, and the
    cache bookkeeping (`get_num_used_shared_caches`, `empty_shared_cache`, `truncate_shared_caches`) must follow."""
    from hydragen_amd.llama import SharedCacheOp

    dtype = torch.float16
    model = make_model(dtype, head_dim=64)
    model.graph(graph)
    g = torch.Generator(device=DEV).manual_seed(17)
    rnd = lambda *s: torch.randint(1, 512, s, device=DEV, generator=g)
    prompt, bs, new = rnd(1, 29), 4, 5
    model.setup_caches(max_unique_batch_size=bs, max_unique_seq_length=32, max_shared_batch_sizes=[1, 2],
                       max_shared_seq_lengths=[29, 11])

    def check(logits, overrides, parts):
        for j in range(bs):
            full = torch.cat(parts(j) + [overrides[j, : new - 1]])[None]
            ref = ref_logits(model, full)[0]
            got = torch.stack([l[j] for l in logits])
            assert (got - ref[full.shape[1] - new :]).abs().max() < 0.75

    # (a) extend, then reuse
    ov1 = rnd(bs, new)
    _, logits = model.generate(input_ids=prompt, num_return_sequences=bs, max_new_tokens=new, temperature=0.0,
                               return_logits=True, shared_cache_op=SharedCacheOp.EXTEND, token_overrides=ov1)
    assert model.get_num_used_shared_caches() == 1
    check(logits, ov1, lambda j: [prompt[0]])
    starting = logits[0][0:1]
    for _ in range(2):
        ov = rnd(bs, new)
        _, lg = model.generate(starting_logits=starting, num_return_sequences=bs, max_new_tokens=new, temperature=0.0,
                               return_logits=True, token_overrides=ov)
        assert model.get_num_used_shared_caches() == 1  # "preserve" keeps the prompt, adds nothing
        check(lg, ov, lambda j: [prompt[0]])
    # a second level on top of the kept prompt, removed again after the call ("preserve")
    lvl2, ov = rnd(2, 11), rnd(bs, new)
    _, lg = model.generate(input_ids=lvl2, num_return_sequences=bs // 2, max_new_tokens=new, temperature=0.0,
                           return_logits=True, token_overrides=ov)
    assert model.get_num_used_shared_caches() == 1
    check(lg, ov, lambda j: [prompt[0], lvl2[j // 2]])
    model.empty_shared_cache()
    assert model.get_num_used_shared_caches() == 0

    # (b) manual cache operations
    prefill = model.append_shared(input_ids=prompt)
    assert model.get_num_used_shared_caches() == 1
    ov = rnd(bs, new)
    _, lg = model.generate(starting_logits=prefill[:, -1], num_return_sequences=bs, max_new_tokens=new, temperature=0.0,
                           return_logits=True, token_overrides=ov)
    check(lg, ov, lambda j: [prompt[0]])
    model.append_shared(input_ids=lvl2)
    assert model.get_num_used_shared_caches() == 2
    model.truncate_shared_caches(1)
    assert model.get_num_used_shared_caches() == 1
    model.empty_shared_cache()
    assert model.get_num_used_shared_caches() == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_layer_glue_equals_the_spelled_out_layers(dtype):
    """HydragenLlamaModel.forward with every residual add fused into the following RMSNorm (hyd_add_rmsnorm) and the SwiGLU
    kernel, against the same weights run layer by layer as llama.py:610-633 spells it (torch add, torch rms_norm, torch
    silu * up): same dataflow, so the hidden states agree to a few roundings of the 16-bit residual stream."""
    from hydragen_amd import layer_ops

    torch.manual_seed(11)
    model = make_model(dtype, layers=3)
    m = model.model
    ids = torch.randint(1, model.config.vocab_size, (5, 9), device=DEV)
    pos = torch.arange(9, device=DEV)[None].expand(5, 9).contiguous()
    model.setup_caches(max_unique_batch_size=5, max_unique_seq_length=32, max_shared_batch_sizes=[], max_shared_seq_lengths=[])
    model.set_mode("unique-prefill")
    with torch.no_grad():
        fused = m(ids, pos)
        supported = layer_ops.supported
        layer_ops.supported = lambda *a, **k: False  # the torch form everywhere
        try:
            model.setup_caches(max_unique_batch_size=5, max_unique_seq_length=32, max_shared_batch_sizes=[], max_shared_seq_lengths=[])
            plain = m(ids, pos)
        finally:
            layer_ops.supported = supported
    assert fused.dtype == dtype and fused.shape == plain.shape
    # measured over seeds (tests/probes/glue_err_probe.py): fp16 relative L2 1.0e-3, max 1.5e-3 of the largest value; bf16 7.9e-3,
    # 1.2e-2 -- a handful of 16-bit roundings taken in a different order; bounds = 2.5 x that
    d = fused.float() - plain.float()
    scale = plain.float().abs().max().item()
    l2 = (d.norm() / plain.float().norm()).item()
    assert l2 <= (2.0e-2 if dtype == torch.bfloat16 else 2.5e-3), l2
    assert d.abs().max().item() <= (3.0e-2 if dtype == torch.bfloat16 else 4.0e-3) * scale, (d.abs().max().item(), scale)


def test_fused_projection_weights_follow_a_replaced_projection():
    """the q | k | v and gate | up GEMMs read one fused weight of which the modules' weights are views.
    Replacing ONE projection later (a quantised / merged k_proj, up_proj) must not leave the stale fused tensor in use, and
    unfuse_weights() gives every projection its own storage back."""
    from hydragen_amd.llama import unfuse_weights

    torch.manual_seed(3)
    model = make_model(torch.bfloat16, layers=1)
    layer = model.model.layers[0]
    x = torch.randn(4, 1, model.config.hidden_size, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        y0 = layer.mlp(x)
        assert layer.mlp._gate_up is not None and layer.mlp.up_proj.weight.data_ptr() != layer.mlp._gate_up.data_ptr()
        assert layer.mlp.gate_proj.weight.data_ptr() == layer.mlp._gate_up.data_ptr()
        layer.mlp.up_proj.weight.data = torch.zeros_like(layer.mlp.up_proj.weight)  # replaced on its own: silu(g) * 0 = 0
        y1 = layer.mlp(x)
        assert float(y1.abs().max()) == 0.0 and float(y0.abs().max()) > 0.0
        assert layer.mlp.up_proj.weight.data_ptr() == layer.mlp._gate_up.data_ptr() + layer.mlp.gate_proj.weight.numel() * 2  # fused again
        unfuse_weights(model)
        assert layer.mlp._gate_up is None
        ptrs = {layer.mlp.gate_proj.weight.untyped_storage().data_ptr(), layer.mlp.up_proj.weight.untyped_storage().data_ptr()}
        assert len(ptrs) == 2
        assert float(layer.mlp(x).abs().max()) == 0.0
