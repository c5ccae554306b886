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
