"""-m "not gpu": the N>1 path (head sharding + all-reduce of the row-parallel o_proj output,
/root/reference/hydragen/tp.py:90-124) with world_size 2 on the gloo backend.  The attention itself
is computed by the CPU oracle here (tests may); on GPUs each rank runs the HIP operator instead."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    from hydragen_amd import tp, utils
    from oracle import hydragen_oracle as O
    from tests.cases import make_case

    assert utils.maybe_init_dist(backend="gloo") == rank
    case = make_case(sizes=[[40], [5, 9, 2, 7]], qheads=8, kvheads=4, dim=64, dtype="f16", seed=3)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    q, k, v = t(case["q"]), t(case["k"]), t(case["v"])
    sks, svs = [t(x) for x in case["shared_ks"]], [t(x) for x in case["shared_vs"]]
    hidden = 8 * 64
    w = torch.from_numpy(np.random.default_rng(0).standard_normal((hidden, hidden)).astype(np.float32)) / 16

    def attn(q_, k_, v_, sk_, sv_):
        return torch.from_numpy(O.hydragen_attention_nopad(q_.numpy(), k_.numpy(), v_.numpy(),
                                                           [x.numpy() for x in sk_], [x.numpy() for x in sv_],
                                                           case["seq_lens"])).float()

    full = attn(q, k, v, sks, svs).reshape(4, 1, hidden) @ w.T
    ql, kl, vl, skl, svl = tp.shard_attention_inputs(q, k, v, sks, svs)
    assert ql.shape[2] == 8 // world and kl.shape[2] == 4 // world
    part = attn(ql, kl, vl, skl, svl).reshape(4, 1, hidden // world) @ tp.shard_o_proj_weight(w).T
    tp.all_reduce_sum(part)
    err = (part - full).abs().max().item()
    if rank == 0:
        ret.put(err)
    dist.barrier()
    dist.destroy_process_group()


def test_head_sharded_attention_plus_allreduce_world2():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) < 1e-4


def test_shard_range_matches_tensor_split():
    from hydragen_amd import tp
    x = torch.arange(32)
    for world in (1, 2, 4, 8):
        for r in range(world):
            assert torch.equal(x[tp.shard_range(32, r, world)], torch.tensor_split(x, world)[r])
    with pytest.raises(AssertionError):
        tp.shard_range(6, 0, 4)


def _tiny_config():
    from hydragen_amd.llama import LlamaConfig

    return LlamaConfig(hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2, vocab_size=50, max_position_embeddings=64, attention_bias=True)


def _tp_model_worker(rank, world, port, shard_dir, ret):
    """Each rank loads its `{rank}.pt` shard, runs the sharded model shell in the reference's "noattention" mode
    (attention replaced by identity on q, llama.py:433-437 -- everything around the HIP operator: column/row
    parallel projections, both all-reduces, replicated embedding / norms / lm_head) and returns logits."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    from hydragen_amd import tp, utils

    assert utils.maybe_init_dist(backend="gloo") == rank
    model = tp.from_pretrained_tp(shard_dir, dtype=torch.float32, device="cpu")
    assert model.config.num_attention_heads == 4 // world and model.config.num_key_value_heads == 2 // world
    assert model.model.layers[0].self_attn.q_proj.weight.shape == (64 // world, 64)
    assert model.model.layers[0].mlp.down_proj.weight.shape == (64, 96 // world)
    model.model.set_disable_attention(True)
    model.model.set_disable_hydragen(True)
    ids = torch.arange(12).reshape(2, 6) % 50
    pos = torch.arange(6).unsqueeze(0).expand(2, -1)
    with torch.no_grad():
        logits = model(ids, pos, full_logits=True)
    if rank == 0:
        ret.put(logits.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_apply_tp_shards_files_and_model_logits_world2(tmp_path):
    """make_tp_files -> from_pretrained_tp -> forward on 2 gloo ranks == the unsharded model (tp.py:115-180)."""
    from hydragen_amd import tp
    from hydragen_amd.llama import HydragenLlamaForCausalLM

    full = HydragenLlamaForCausalLM.from_config(_tiny_config(), dtype=torch.float32, device="cpu", seed=5, std=0.2)
    with torch.no_grad():
        for lyr in full.model.layers:  # biases too (rowwise bias must be added exactly once)
            for lin in (lyr.self_attn.q_proj, lyr.self_attn.k_proj, lyr.self_attn.v_proj, lyr.self_attn.o_proj):
                lin.bias.normal_(0.0, 0.2)
    tp.make_tp_files(full, tmp_path, num_splits=2)
    assert sorted(f.name for f in tmp_path.glob("*.pt")) == ["0.pt", "1.pt"]
    full.model.set_disable_attention(True)
    full.model.set_disable_hydragen(True)
    ids = torch.arange(12).reshape(2, 6) % 50
    pos = torch.arange(6).unsqueeze(0).expand(2, -1)
    with torch.no_grad():
        want = full(ids, pos, full_logits=True).numpy()

    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_model_worker, args=(r, 2, port, str(tmp_path), ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = ret.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())


def test_apply_tp_world1_is_identity_and_rejects_indivisible_heads():
    from hydragen_amd import tp
    from hydragen_amd.llama import HydragenLlamaForCausalLM

    m = HydragenLlamaForCausalLM.from_config(_tiny_config(), dtype=torch.float32, device="cpu")
    tp.apply_tp(m, rank=0, world_size=1)
    assert m.config.num_attention_heads == 4 and not m.model.layers[0].mlp.tp_reduce
    with pytest.raises(AssertionError):
        tp.apply_tp(m, rank=0, world_size=4)  # 2 kv heads do not divide over 4 ranks
