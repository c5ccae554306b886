"""
Host side of `hyd_allreduce_sum` (include/hydragen_hip.h): the all-reduce(sum) behind the row-parallel o_proj /
down_proj of /root/reference/hydragen/tp.py:83-87,108-112 as a two-shot direct exchange over peer-mapped device
memory (xGMI), instead of a RCCL ring.  One process per GPU; `torch.distributed` (any backend) is used once, to
exchange the IPC handles of the per-rank shared blocks.

    comm = XgmiAllReduce(max_bytes=32 << 20)      # after init_process_group; collective
    comm.all_reduce_(x)                           # in place, on torch's current stream; HIP-graph capturable
    hydragen_amd.tp.use_xgmi_allreduce(comm)      # route tp.all_reduce_sum through it

The shared block is a raw, uncached HIP allocation (hipIpcGetMemHandle needs an allocation base, which a tensor of
the caching allocator is not); it is released in `close()`.
"""

from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist
from torch import Tensor

from . import _lib
from ._lib import HYD_BF16, HYD_F16, HYD_F32, AllReduceParams

_DT = {torch.float16: HYD_F16, torch.bfloat16: HYD_BF16, torch.float32: HYD_F32}
_hip = None


def _runtime():
    """The HIP runtime torch already mapped (by its path in /proc/self/maps: a second copy of libamdhip64 found through
    the loader's search path would be a second runtime instance in this process)."""
    global _hip
    if _hip is None:
        import torch  # noqa: F401
        path = "libamdhip64.so"
        try:
            for line in open("/proc/self/maps"):
                if "libamdhip64.so" in line:
                    path = line.split()[-1]
                    break
        except OSError:
            pass
        _hip = C.CDLL(path)
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip


_HIP_DEVICE_MALLOC_UNCACHED = 0x3
_HIP_DEVICE_MALLOC_FINEGRAINED = 0x1


class XgmiAllReduce:
    def __init__(self, max_bytes: int, group=None, device: torch.device | None = None, timeout_log2_polls: int = 0):
        """timeout_log2_polls: every wait on a peer gives up after 2^n polls of ~0.3 us (0 = the library default, 2^27,
        ~40 s).  A rank that gives up leaves garbage in its output and sets the block's status word: call `check()`
        (or `hydragen_amd.tp.check_collectives()`) before trusting the results of a run."""
        assert dist.is_initialized(), "init_process_group first: the IPC handles are exchanged through it"
        self.lib = _lib.load()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 8:
            raise NotImplementedError("one node: at most 8 ranks")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.max_bytes = int(max_bytes)
        self.timeout_log2_polls = int(timeout_log2_polls)
        self.block_bytes = self.lib.hyd_allreduce_block_bytes(self.world, self.max_bytes)
        assert self.block_bytes > 0
        hip = _runtime()
        with torch.cuda.device(self.device):
            own = C.c_void_p()
            # Uncached (else fine-grained) device memory: peers write the block's flags and read its staged data over
            # the fabric, which does not probe the owner's L2 for ordinary coarse-grained allocations.
            self.uncached = hip.hipExtMallocWithFlags(C.byref(own), self.block_bytes, _HIP_DEVICE_MALLOC_UNCACHED) == 0
            if not self.uncached:
                if hip.hipExtMallocWithFlags(C.byref(own), self.block_bytes, _HIP_DEVICE_MALLOC_FINEGRAINED) != 0:
                    raise RuntimeError("no uncached / fine-grained device allocation for the all-reduce block")
            hip.hipMemset(own, 0, self.block_bytes)
            torch.cuda.synchronize()
            handle = (C.c_ubyte * 64)()
            _lib.check(self.lib.hyd_ipc_get_handle(own, handle))
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle), group=group)
            self._own = own
            self._opened = []
            ptrs = []
            for r, h in enumerate(handles):
                if r == self.rank:
                    ptrs.append(own.value)
                    continue
                p = C.c_void_p()
                buf = (C.c_ubyte * 64).from_buffer_copy(h)
                _lib.check(self.lib.hyd_ipc_open_handle(buf, C.byref(p)))
                self._opened.append(p)
                ptrs.append(p.value)
            self._blocks = (C.c_void_p * self.world)(*ptrs)
        dist.barrier(group=group)  # every block is mapped everywhere before the first call

    def all_reduce_(self, x: Tensor) -> Tensor:
        """In-place sum over the ranks; x: contiguous fp16 / bf16 / fp32 CUDA tensor of at most max_bytes."""
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("all_reduce_ takes a contiguous CUDA tensor")
        p = AllReduceParams()
        p.blocks = C.cast(self._blocks, C.POINTER(C.c_void_p))
        p.in_, p.out, p.count = x.data_ptr(), x.data_ptr(), x.numel()
        p.max_bytes, p.dtype, p.rank, p.world = self.max_bytes, _DT[x.dtype], self.rank, self.world
        p.timeout_log2_polls = self.timeout_log2_polls
        _lib.check(self.lib.hyd_allreduce_sum(C.byref(p), torch.cuda.current_stream().cuda_stream))
        return x

    def status(self) -> int:
        """0 = every call so far completed; 1 / 2 = a peer did not show up in shot 1 / 2 (synchronises)."""
        torch.cuda.synchronize()
        word = C.c_uint32()
        _runtime().hipMemcpy(C.byref(word), self.lib.hyd_allreduce_status(self._own), 4, 2)  # device -> host
        return int(word.value)

    def check(self) -> None:
        """Raise if any call so far gave up on a peer (its output was not the sum).  Synchronises the device."""
        st = self.status()
        if st:
            raise RuntimeError(f"xGMI all-reduce: rank {self.rank} gave up waiting for a peer in shot {st} "
                               f"(2^{self.timeout_log2_polls or 27} polls); results of this run are invalid")

    def close(self):
        if getattr(self, "_own", None) is None:
            return
        torch.cuda.synchronize()
        for p in self._opened:
            self.lib.hyd_ipc_close_handle(p)
        _runtime().hipFree(self._own)
        self._own, self._opened = None, []

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
