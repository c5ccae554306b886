"""-m gpu: the direct xGMI all-reduce (`hyd_allreduce_sum`, replaces the NCCL all-reduce of
/root/reference/hydragen/tp.py:83-87,108-112) with TWO processes sharing the one GPU of the box: each rank maps the
other's block through hipIpc handles exactly as it would across two GPUs, so the protocol (staging, epochs, flags,
system-scope fences, slice ownership, tails, repeated calls, graph replay) is exercised end to end; only the link
bandwidth needs a multi-GPU node.  Reference: the sum of the ranks' inputs computed on the host."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        from hydragen_amd.xgmi_allreduce import XgmiAllReduce

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = XgmiAllReduce(max_bytes=(33 << 20) if world > 2 else (9 << 20))
        worst = 0.0
        cases = [(torch.bfloat16, 1024 * 4096), (torch.float16, 4097), (torch.float32, 1000003), (torch.bfloat16, 8),
                 (torch.bfloat16, 7), (torch.float16, 2 * 1024 * 1024 + 24)]
        if world > 2:
            # the 8-slice exchange of the C5 message (32 MiB bf16: [2048, 1, 8192], tp.py:108-112), counts below world x 8
            # (empty slices), odd counts (a ragged last slice), fp32 at a size where every workgroup of shot 2 has work
            cases += [(torch.bfloat16, 2048 * 8192), (torch.bfloat16, world * 8 - 3), (torch.float16, world + 1),
                      (torch.float32, 3), (torch.bfloat16, 8 * 1024 * 1024 + 13), (torch.float32, 4 * 1024 * 1024 + 5)]
        for rep in range(3 if world <= 2 else 2):  # repeated calls: epochs, flag reuse, staging reuse
            for dt, n in cases:
                g = torch.Generator(device="cuda").manual_seed(100 * rep + n % 997 + rank)
                x = torch.randn(n, device="cuda", dtype=dt, generator=g)
                parts = [torch.empty(n, dtype=dt) for _ in range(world)]
                dist.all_gather(parts, x.cpu())
                want = sum(p.float() for p in parts)
                comm.all_reduce_(x)
                torch.cuda.synchronize()
                tol = 1e-6 if dt == torch.float32 else (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -10)
                err = ((x.float().cpu() - want).abs() / (want.abs() + 1.0)).max().item()
                assert err <= tol * 2, (rank, rep, dt, n, err)
                worst = max(worst, err)
        assert comm.status() == 0
        # HIP-graph capture + replay with changing inputs (llama.py:849-854 captures the all-reduce with the forward)
        x = torch.zeros(1024, 1, 4096, device="cuda", dtype=torch.bfloat16)
        src = torch.empty_like(x)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                x.copy_(src.normal_())
                comm.all_reduce_(x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        dist.barrier()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            x.copy_(src)
            comm.all_reduce_(x)
        for rep in range(3):
            src.normal_(generator=torch.Generator(device="cuda").manual_seed(7 * rep + rank))
            torch.cuda.synchronize()
            parts = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)]
            dist.all_gather(parts, src.cpu())
            want = sum(p.float() for p in parts)
            graph.replay()
            torch.cuda.synchronize()
            err = ((x.float().cpu() - want).abs() / (want.abs() + 1.0)).max().item()
            assert err <= 2.0 ** -7, (rank, "graph", rep, err)
        assert comm.status() == 0
        dist.barrier()
        unc = comm.uncached
        comm.close()
        dist.destroy_process_group()
        ret.put((rank, "ok", (worst, unc)))
    except Exception as e:  # report instead of hanging the parent
        import traceback
        ret.put((rank, "fail", traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_xgmi_allreduce_ranks_on_one_gpu(world):
    """world = 4 and 8 rehearse what a real node runs (tp.py:115-132; the reference's sweeps use --nproc_per_node=8,
    docs/sweeps_from_paper.md:25-150): the 8-slice two-shot exchange, peers dealt over the workgroups of shot 2, every
    rank's workgroups co-resident on the one device while they wait for each other."""
    import torch.multiprocessing as mp

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    res = [ret.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, state, info in res:
        assert state == "ok", f"rank {rank}: {info}"
    print("worst error, uncached block:", [info for _, _, info in res])


def test_missing_peer_times_out_with_status():
    """A peer that never shows up: the call must finish (bounded polls), leave the input untouched and report it in
    the status word -- not hang the stream.  The 'peer' is a second, silent block in this process."""
    import ctypes as C
    import time

    from hydragen_amd import _lib
    from hydragen_amd._lib import HYD_BF16, AllReduceParams

    lib = _lib.load()
    max_bytes = 1 << 20
    nbytes = lib.hyd_allreduce_block_bytes(2, max_bytes)
    blocks_t = [torch.zeros(nbytes, device="cuda", dtype=torch.uint8) for _ in range(2)]
    blocks = (C.c_void_p * 2)(*[b.data_ptr() for b in blocks_t])
    x = torch.randn(4096, device="cuda", dtype=torch.bfloat16)
    x0 = x.clone()
    p = AllReduceParams()
    p.blocks = C.cast(blocks, C.POINTER(C.c_void_p))
    p.in_, p.out, p.count = x.data_ptr(), x.data_ptr(), x.numel()
    p.max_bytes, p.dtype, p.rank, p.world = max_bytes, HYD_BF16, 0, 2
    p.timeout_log2_polls = 22  # ~1.3 s per wait instead of the default ~40 s
    t0 = time.time()
    _lib.check(lib.hyd_allreduce_sum(C.byref(p), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    dt = time.time() - t0
    word = torch.empty(1, dtype=torch.int32)
    status_ptr = lib.hyd_allreduce_status(C.c_void_p(blocks_t[0].data_ptr()))
    off = (status_ptr if isinstance(status_ptr, int) else C.cast(status_ptr, C.c_void_p).value) - blocks_t[0].data_ptr()
    status = int(blocks_t[0][off:off + 4].view(torch.int32).item())
    print(f"missing peer: returned after {dt:.2f} s with status {status}")
    assert status != 0
    assert dt < 60
    assert torch.equal(x, x0)


def _rccl_worker(port, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist

        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)  # "nccl" IS RCCL on ROCm
        x = torch.randn(1024, 4096, device="cuda", dtype=torch.bfloat16)
        want = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert torch.equal(x, want)
        # the reference's in-graph collective (llama.py:849-854): the all-reduce of a [B, 1, hidden] block output captured
        # behind a kernel of ours and replayed
        from hydragen_amd import layer_ops
        w = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                y = layer_ops.add_rms_norm(want, None, w, 1e-5)[1]
                dist.all_reduce(y, op=dist.ReduceOp.SUM)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = layer_ops.add_rms_norm(want, None, w, 1e-5)[1]
            dist.all_reduce(y, op=dist.ReduceOp.SUM)
        ref = layer_ops.add_rms_norm(want, None, w, 1e-5)[1].clone()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
        ret["ok"] = True
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        ret["err"] = repr(e)


def test_rccl_initialises_reduces_and_captures_with_one_rank():
    """One GPU allows one RCCL rank (RCCL refuses two ranks on a device): at least the library loads, builds a communicator,
    runs its all-reduce kernel and lets it be captured into a HIP graph next to our kernels -- the multi-rank run needs a
    multi-GPU node (tp.py:83-87,108-112; llama.py:849-854)."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_rccl_worker, args=(port, ret))
    p.start()
    p.join(180)
    if p.is_alive():
        p.kill()
        pytest.fail("RCCL single-rank worker hung")
    assert ret.get("ok"), ret.get("err")
