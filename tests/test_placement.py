"""hydragen_amd/placement.py: the unique K|V caches are allocated where the suffix pass streams them fastest.
CPU: the plan and the choice; GPU: candidates are allocated, timed, the fastest kept and zeroed, the rest released."""
import pytest
import torch

from hydragen_amd import placement as P

GIB = 1 << 30


def test_plan_one_arena_spreads_candidates_over_the_span_and_respects_free_memory():
    n, spacer = P.plan(1, 2 * GIB, 250 * GIB, 6)
    assert n == 6 and spacer % (2 << 20) == 0
    assert abs(n * 2 * GIB + (n - 1) * spacer - P.SPAN_GIB * GIB) < 16 << 20          # candidates + spacers = the span
    n, spacer = P.plan(1, 2 * GIB, 20 * GIB, 6)                                         # 10 GiB may be used: 5 candidates, no room for spacers
    assert n == 5 and n * 2 * GIB + (n - 1) * spacer <= 10 * GIB
    assert P.plan(1, 2 * GIB, 6 * GIB, 6) == (1, 0)                                    # not even two candidates fit
    assert P.plan(1, 64 << 20, 250 * GIB, 6) == (1, 0)                                 # lives in the memory-side cache: nothing to choose
    assert P.plan(1, 2 * GIB, 250 * GIB, 1) == (1, 0)                                  # switched off


def test_plan_many_arenas_over_allocates_by_a_fraction():
    assert P.plan(32, 2 * GIB, 250 * GIB, 6) == (48, 0)
    assert P.plan(32, 2 * GIB, 140 * GIB, 6) == (35, 0)                                # 70 GiB usable: 3 extra
    assert P.plan(32, 2 * GIB, 100 * GIB, 6) == (32, 0)                                # no room: plain allocation


def test_choose_keeps_the_fastest_in_allocation_order():
    assert P.choose([5.0, 3.0, 4.0, 1.0], 2) == [1, 3]
    assert P.choose([2.0, 2.0, 2.0], 2) == [0, 1]
    assert P.choose([9.0], 1) == [0]


def test_cpu_allocation_is_plain_zeros():
    arenas, rep = P.place_kv_arenas(3, (2, 16, 1, 64), torch.float16, "cpu", 4)
    assert len(arenas) == 3 and all(a.shape == (2, 2, 16, 1, 64) and not a.any() for a in arenas)
    assert rep == {"candidates": 3, "probed": False, "why": "not a GPU allocation"}


def test_kv_arena_puts_a_sequences_k_and_v_side_by_side():
    """Logical [2, B, rows, H, D] (arena[0] / arena[1] = the reference's two caches, llama.py:186-198), memory [B, 2, rows, H, D]:
    a sequence's V rows follow its K rows, consecutive sequences sit 2 x rows token rows apart, heads and dims stay contiguous."""
    b, rows, h, d = 3, 16, 2, 64
    arena = P.kv_arena((b, rows, h, d), torch.float16, "cpu")
    assert arena.shape == (2, b, rows, h, d) and not arena.any()
    k, v = arena[0], arena[1]
    tok = h * d
    assert k.stride() == (2 * rows * tok, tok, d, 1) == v.stride()
    assert v.data_ptr() - k.data_ptr() == rows * tok * arena.element_size()           # V of sequence 0 right behind its K
    assert k[1].data_ptr() - k[0].data_ptr() == 2 * rows * tok * arena.element_size()  # next sequence: 2 x rows further
    k[1, 5, 1, 7] = 3.0  # the views alias one allocation
    assert arena.permute(1, 0, 2, 3, 4).is_contiguous() and arena.permute(1, 0, 2, 3, 4)[1, 0, 5, 1, 7] == 3.0


def test_set_candidates_returns_the_previous_setting():
    old = P.set_candidates(1)
    try:
        assert P.get_candidates() == 1 and P.set_candidates(9) == 1 and P.get_candidates() == 9
    finally:
        P.set_candidates(old)


@pytest.mark.gpu
def test_fastest_candidates_are_kept_zeroed_and_the_rest_released(monkeypatch):
    monkeypatch.setattr(P, "MIN_ARENA_BYTES", 1 << 20)
    monkeypatch.setattr(P, "SPAN_GIB", 0.25)
    old = P.set_candidates(5)
    try:
        torch.cuda.empty_cache()
        before = torch.cuda.memory_reserved()
        seen = []

        def probe(a):  # candidates arrive one by one, all alive at once (distinct memory); the third is "fastest"
            a.fill_(1)
            seen.append(a.data_ptr())
            return 100.0 - (30.0 if len(seen) == 3 else 0.0) - len(seen)

        (arena,), rep = P.place_kv_arenas(1, (256, 128, 2, 128), torch.bfloat16, "cuda:0", 8, probe=probe)  # 32 MiB each: own segments
        assert rep["probed"] and rep["candidates"] == 5 and rep["kept"] == [2] and len(set(seen)) == 5
        assert arena.data_ptr() == seen[2] and arena.shape == (2, 256, 128, 2, 128) and not arena.any()
        # what was not kept went back to the driver (torch's small-block segments aside)
        assert torch.cuda.memory_reserved() - before <= arena.numel() * 2 + (24 << 20)
        # many arenas: over-allocation by half, the slowest dropped, allocation order kept
        seen.clear()
        arenas, rep = P.place_kv_arenas(4, (64, 128, 2, 128), torch.bfloat16, "cuda:0", 8, probe=lambda a: [seen.append(a.data_ptr()), float(len(seen) % 3)][1])
        assert rep["candidates"] == 6 and rep["kept"] == [0, 2, 3, 5] and [a.data_ptr() for a in arenas] == [seen[i] for i in rep["kept"]]
        # a probe that REFUSES THE SHAPE (the C ABI's bad-argument / unsupported codes, the mirrors' asserts) leaves plain caches
        # behind, with a warning, not an exception ...
        def refused(a):
            raise NotImplementedError("hydragen_hip: head_dim 96: only 64, 128 and 256 are implemented")
        with pytest.warns(UserWarning, match="refused this cache shape"):
            arenas, rep = P.place_kv_arenas(2, (64, 128, 2, 128), torch.bfloat16, "cuda:0", 8, probe=refused)
        assert len(arenas) == 2 and not rep["probed"] and "head_dim 96" in rep["why"] and not any(a.any() for a in arenas)
        # ... but a runtime fault (launch failure, HIP error) is raised here, next to its cause
        def broken(a):
            raise RuntimeError("hydragen_hip error -4: suffix kernel launch failed: hip error 719")
        with pytest.raises(RuntimeError, match="launch failed"):
            P.place_kv_arenas(2, (64, 128, 2, 128), torch.bfloat16, "cuda:0", 8, probe=broken)
    finally:
        P.set_candidates(old)


@pytest.mark.gpu
def test_the_real_probe_times_a_suffix_pass_and_leaves_the_arena_alone():
    a = torch.randn(2, 256, 64, 4, 128, device="cuda:0", dtype=torch.bfloat16)
    ref = a.clone()
    us = P.probe_suffix_pass_us(a, 8)
    assert 1.0 < us < 1e4 and torch.equal(a, ref)
    old = P.set_candidates(1)
    try:
        (arena,), rep = P.place_kv_arenas(1, (2048, 128, 8, 128), torch.bfloat16, "cuda:0", 8)
        assert rep == {"candidates": 1, "probed": False, "why": "probing off"} and not arena.any()
    finally:
        P.set_candidates(old)


def test_longest_first_schedule_and_its_entry_check():
    """`flash.longest_first` / `flash.seq_order` (the schedule hint behind hyd_suffix_params.seq_order): host logic, no GPU needed."""
    from hydragen_amd.flash import current_seq_order, longest_first, seq_order

    lens = torch.tensor([5, 9, 5, 0, 12, 9], dtype=torch.int32)
    order = longest_first(lens)
    assert order.dtype == torch.int32 and order.tolist() == [4, 1, 5, 0, 2, 3]  # stable: equal lengths keep their index order
    assert current_seq_order() is None
    with seq_order(order):
        assert current_seq_order() is order
        with seq_order(None):
            assert current_seq_order() is None
        assert current_seq_order() is order
    assert current_seq_order() is None
    with pytest.raises(ValueError, match="not a permutation"):
        seq_order(torch.tensor([0, 1, 1], dtype=torch.int32))
    with pytest.raises(ValueError, match="int32"):
        seq_order(torch.tensor([0, 1, 2]))
    seq_order(torch.tensor([0, 1, 1], dtype=torch.int32), check=False)  # the caller may vouch for it (the model shell does)
