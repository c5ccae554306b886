"""
Where a sequence-private K/V cache sits in HBM decides the rate the suffix pass can stream it at.

Measured on MI355X (profiles/r05_gqa_placement.md, addendum): 48 successive 2 GiB allocations of one process, the same
`hyd_suffix_attn_fwd` launch timed on each, fall into runs of 5-13 consecutive allocations (10-26 GiB) at two rates --
6.8 TB/s and 6.3 TB/s (8 % apart for the grouped-query kernel, 4 % for the dot-product kernel).  The rate belongs to the
allocation: it is the same on every stream, for any data, on a second pass, and the FIRST large allocation of a process
is in a slow run in most starts.  The caches are this package's allocations (the reference allocates them in
`/root/reference/hydragen/llama.py:186-198`; `setup_caches`, `llama.py:930-960`), so the package picks: more candidates
than needed are allocated at once, each is timed with the suffix pass that will serve it (half and all of its keys, at
the memory system's limit), the fastest are kept, the rest go back to the driver.

Nothing here is on the data path: it runs once per `setup_caches` (or once per bench process), before any graph capture.
`set_candidates(1)` / `HYDRAGEN_KV_CANDIDATES=1` turns it off (plain `torch.zeros`, the reference's behaviour).
"""

from __future__ import annotations

import math
import os
from typing import Callable, Sequence

import torch
from torch import Tensor

# candidates tried for ONE arena (spaced out over `SPAN_GIB` of fresh memory so that they do not all sit in one run);
# a request for many arenas (a model's layers) over-allocates by EXTRA_FRACTION instead (they span several runs anyway)
_candidates = int(os.environ.get("HYDRAGEN_KV_CANDIDATES", "6"))
SPAN_GIB = 48.0
EXTRA_FRACTION = 0.5
MIN_ARENA_BYTES = 512 << 20   # at or near the 256 MB memory-side cache's size placement does not show (a 268 MB arena, the C5 TP = 8
                              # slice: six candidates within 2 %, nothing gained -- profiles/r05_kv_placement_probe.md); measured gains are at 2 GiB
MAX_FREE_FRACTION = 0.5       # of the device's free memory, the most the candidates + spacers may take transiently


def kv_arena(shape: Sequence[int], dtype: torch.dtype, device, zero: bool = True) -> Tensor:
    """One layer's unique K|V cache, logical shape [2 (K | V), batch, rows, kv heads, head dim] -- `arena[0]` / `arena[1]` are the
    reference's two caches (`/root/reference/hydragen/llama.py:186-198`) -- laid out in memory as [batch, 2, rows, kv heads, head dim]:
    a sequence's K rows, then its V rows.  Same bytes, but consecutive sequences sit 2 x rows token rows apart instead of `rows`, and
    the suffix pass streams a 128-row cache 1.5-3 % faster that way (both kernels, three fresh allocations of each layout,
    alternating: 167-169 -> 164-167 us at S = 64, 327-330 -> 320-325 at S = 128, profiles/r06_kv_arena_layout_ab.txt; the rate grows
    with the distance between sequences up to 512 rows, profiles/r06_suffix_rows_capacity.txt).  The operators take strided K / V."""
    b, rows, hkv, d = (int(x) for x in shape)
    make = torch.zeros if zero else torch.empty
    return make((b, 2, rows, hkv, d), dtype=dtype, device=device).permute(1, 0, 2, 3, 4)


def set_candidates(n: int) -> int:
    """Candidates per arena (1 = no probing).  Returns the previous setting."""
    global _candidates
    old, _candidates = _candidates, max(1, int(n))
    return old


def get_candidates() -> int:
    return _candidates


def probe_suffix_pass_us(arena: Tensor, qheads: int, iters: int = 3) -> float:
    """Microseconds the suffix pass that will serve this cache takes on it: `flash_attention_seqlen` (= `hyd_suffix_attn_fwd`,
    the kernel the library picks for `qheads` query heads) over `arena` = [2 (K | V), batch, rows, kv heads, head dim], at
    HALF of the cache's rows per sequence and at all of them, min of `iters` launches each after a warm-up, summed (HIP
    events on the current stream).  The real kernel on the real strides, because the effect is one of placement x access
    pattern: a stand-in (the one-row decode kernel over a grouped-query cache) ranks the candidates differently
    (profiles/r05_kv_placement_probe.md).  Half length = the mean decode step of a generation that fills the cache.  The
    arena's contents do not matter to the rate (measured: random, zeros, stale: +- 1 us of 170) and are not touched."""
    from .flash import flash_attention_seqlen

    _, B, S, Hkv, D = arena.shape
    q = torch.zeros((B, 1, qheads, D), dtype=arena.dtype, device=arena.device)
    total = 0.0
    for rows in sorted({max(1, S // 2), S}):
        lens = torch.full((B,), rows, dtype=torch.int32, device=arena.device)
        best = math.inf
        for i in range(iters + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            flash_attention_seqlen(q, arena[0], arena[1], lens)
            e1.record()
            e1.synchronize()
            if i:
                best = min(best, e0.elapsed_time(e1) * 1e3)
        total += best
    return total


def choose(times_us: Sequence[float], count: int) -> list[int]:
    """Indexes of the `count` fastest candidates, in allocation order (layer i gets the i-th kept arena)."""
    order = sorted(range(len(times_us)), key=lambda i: (times_us[i], i))[:count]
    return sorted(order)


def plan(count: int, arena_bytes: int, free_bytes: int, candidates: int) -> tuple[int, int]:
    """(number of candidates, spacer bytes between them) for `count` arenas of `arena_bytes` with `free_bytes` free on the
    device.  (count, 0) = no probing."""
    if candidates <= 1 or arena_bytes < MIN_ARENA_BYTES:
        return count, 0
    budget = int(free_bytes * MAX_FREE_FRACTION)
    if count == 1:
        n = candidates
        while n > 1 and n * arena_bytes > budget:
            n -= 1
        if n <= 1:
            return 1, 0
        spacer = int((SPAN_GIB * (1 << 30) - n * arena_bytes) / (n - 1))
        spacer = max(0, min(spacer, (budget - n * arena_bytes) // (n - 1)))
        return n, spacer >> 21 << 21
    extra = min(int(math.ceil(count * EXTRA_FRACTION)), max(0, (budget - count * arena_bytes) // arena_bytes))
    return count + int(extra), 0


def place_kv_arenas(count: int, shape: Sequence[int], dtype: torch.dtype, device, qheads: int, *, zero: bool = True,
                    probe: Callable[[Tensor], float] | None = None) -> tuple[list[Tensor], dict]:
    """`count` K|V arenas of logical shape [2, *shape] (shape = [batch, rows, kv heads, head dim]; memory layout: `kv_arena`) on
    `device`, each the unique cache of one layer, placed where the suffix pass streams fastest among the candidates tried.  Returns (arenas,
    report); report = {"candidates", "spacer_bytes", "probe_us", "kept"} -- or {"candidates": count, "probed": False, "why"}.
    `probe(arena) -> us` defaults to `probe_suffix_pass_us(arena, qheads)` (qheads = the query heads that will attend to it).
    Side effects, once per call: up to MAX_FREE_FRACTION (half) of the device's FREE memory is allocated for a moment
    (candidates + spacers) and `torch.cuda.empty_cache()` runs before returning; ranks that share one device should place
    their caches one after the other (they would race on `mem_get_info`).  Only a shape the suffix pass refuses (ValueError /
    NotImplementedError / AssertionError from the operator) falls back to plain allocation, with a warning; runtime faults propagate."""
    dev = torch.device(device)
    full = (2,) + tuple(int(x) for x in shape)
    arena_bytes = math.prod(full) * torch.empty((), dtype=dtype).element_size()

    def plain(why: str):
        return [kv_arena(shape, dtype, dev, zero) for _ in range(count)], {"candidates": count, "probed": False, "why": why}

    if dev.type != "cuda":
        return plain("not a GPU allocation")
    if torch.cuda.is_current_stream_capturing():
        return plain("inside a graph capture")
    free_bytes, _ = torch.cuda.mem_get_info(dev)
    n, spacer = plan(count, arena_bytes, free_bytes, _candidates)
    if n <= count:
        return plain("probing off" if _candidates <= 1 else
                     "arena below the memory-side cache's size" if arena_bytes < MIN_ARENA_BYTES else "not enough free memory for candidates")
    if probe is None:
        probe = lambda a: probe_suffix_pass_us(a, qheads)  # noqa: E731
    cands, spacers = [], []
    try:
        for i in range(n):
            cands.append(kv_arena(shape, dtype, dev, zero=False))
            if spacer and i + 1 < n:
                spacers.append(torch.empty((spacer,), dtype=torch.uint8, device=dev))
    except torch.cuda.OutOfMemoryError:
        if len(cands) < count:
            del cands, spacers
            torch.cuda.empty_cache()
            return plain("out of memory while allocating candidates")
    with torch.cuda.device(dev):
        try:
            times = [float(probe(c)) for c in cands]
        except (ValueError, NotImplementedError, AssertionError) as ex:
            # a SHAPE the suffix pass refuses (bad-argument / unsupported codes of the C ABI, the mirrors' asserts) must not
            # cost the caches: first come, first kept.  Anything else -- a launch failure, a HIP runtime error, a library that
            # does not load -- is a real fault and propagates from here, next to its cause, not from a later graph capture.
            import warnings

            warnings.warn(f"KV placement probe refused this cache shape ({type(ex).__name__}: {str(ex)[:160]}); caches are not placed")
            out = cands[:count]
            del cands, spacers
            torch.cuda.empty_cache()
            if zero:
                for a in out:
                    a.zero_()
            return out, {"candidates": count, "probed": False, "why": f"probe refused the shape: {type(ex).__name__}: {str(ex)[:120]}"}
        kept = choose(times, count)
        out = [cands[i] for i in kept]
        del cands, spacers
        torch.cuda.empty_cache()  # candidates not kept and the spacers go back to the driver, not into torch's pool
        if zero:
            for a in out:
                a.zero_()
    return out, {"candidates": len(times), "probed": True, "spacer_bytes": spacer, "probe_us": [round(t, 1) for t in times], "kept": kept}
