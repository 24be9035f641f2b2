"""marl_reward_events_noalloc (partial sums in the library's static device scratch) against
marl_reward_events (caller-supplied workspace): identical totals, called through the C ABI, two
calls in a row with the second on a smaller input (the scratch keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_POSSIBLE = np.array([float(np.float32(own + k)) for own in (0.0, 0.1, 0.3) for k in range(0, 5)], dtype=np.float32)


def _rewards(rng, n):
    p = np.full(_POSSIBLE.size, 0.03)
    p[0] = 1.0 - 0.03 * (_POSSIBLE.size - 1)
    return rng.choice(_POSSIBLE, size=n, p=p).astype(np.float32)


def _call_pair(L, check, r_host, E, A):
    """one step's rewards through both entry points; returns (totals, first_rows) of each"""
    vp = ctypes.c_void_p
    dev = torch.device("cuda")
    r = torch.from_numpy(r_host).to(dev)
    stream = vp(torch.cuda.current_stream(dev).cuda_stream)
    out = []
    for noalloc in (False, True):
        totals = torch.zeros(4, dtype=torch.int64, device=dev)
        first = torch.zeros(4, dtype=torch.float64, device=dev)
        if noalloc:
            check(L.marl_reward_events_noalloc(vp(r.data_ptr()), E, A, vp(totals.data_ptr()), vp(first.data_ptr()), stream),
                  "marl_reward_events_noalloc")
        else:
            ws = torch.empty(max(int(L.marl_reward_events_workspace_bytes(E, A)), 8) // 8, dtype=torch.int64, device=dev)
            check(L.marl_reward_events(vp(r.data_ptr()), E, A, vp(totals.data_ptr()), vp(first.data_ptr()),
                                       vp(ws.data_ptr()), stream), "marl_reward_events")
        torch.cuda.synchronize()
        out.append((totals.cpu().numpy(), first.cpu().numpy()))
    return out


def test_noalloc_matches_workspace_variant_twice_with_a_smaller_second_input(oracle_mod):
    from marl_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(11)
    # large first (256 chunks of partial sums), then smaller ones that reuse the front of the scratch
    for E, A in [(300001, 2), (5000, 4), (257, 2), (3, 1)]:
        r = _rewards(rng, E * A)
        (t_ws, f_ws), (t_na, f_na) = _call_pair(L, _lib.check, r, E, A)
        np.testing.assert_array_equal(t_na, t_ws)
        np.testing.assert_array_equal(f_na, f_ws)
        t_or, f_or = oracle_mod.reward_events(r, E, A)
        np.testing.assert_array_equal(t_na, t_or)
        np.testing.assert_array_equal(f_na, f_or)


def test_noalloc_accumulates_like_the_workspace_variant():
    from marl_amd import _lib

    L = _lib.lib()
    vp = ctypes.c_void_p
    dev = torch.device("cuda")
    rng = np.random.default_rng(5)
    E, A = 40000, 2
    tot = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(2)]
    fst = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(2)]
    ws = torch.empty(max(int(L.marl_reward_events_workspace_bytes(E, A)), 8) // 8, dtype=torch.int64, device=dev)
    stream = vp(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(6):
        r = torch.from_numpy(_rewards(rng, E * A)).to(dev)
        _lib.check(L.marl_reward_events(vp(r.data_ptr()), E, A, vp(tot[0].data_ptr()), vp(fst[0].data_ptr()),
                                        vp(ws.data_ptr()), stream), "marl_reward_events")
        _lib.check(L.marl_reward_events_noalloc(vp(r.data_ptr()), E, A, vp(tot[1].data_ptr()), vp(fst[1].data_ptr()),
                                                stream), "marl_reward_events_noalloc")
    torch.cuda.synchronize()
    assert torch.equal(tot[0], tot[1]) and torch.equal(fst[0], fst[1])


def test_noalloc_rejects_more_rows_than_its_limit():
    from marl_amd import _lib

    L = _lib.lib()
    r = torch.zeros(8, dtype=torch.float32, device="cuda")
    t = torch.zeros(4, dtype=torch.int64, device="cuda")
    rc = L.marl_reward_events_noalloc(ctypes.c_void_p(r.data_ptr()), (1 << 24) + 1, 1, ctypes.c_void_p(t.data_ptr()), None, None)
    assert rc == -1 and b"at most" in L.marl_last_error()


def test_noalloc_really_allocates_nothing_and_can_be_captured():
    """What the entry point is for: after a first call on the device (which looks up the static
    scratch's address) further calls leave the device's free memory untouched, and a call can be
    captured into a hipGraph and replayed, every replay adding the step's events again."""
    from marl_amd import _lib

    L = _lib.lib()
    vp = ctypes.c_void_p
    dev = torch.device("cuda")
    rng = np.random.default_rng(9)
    E, A = 70000, 2
    r = torch.from_numpy(_rewards(rng, E * A)).to(dev)
    tot = torch.zeros(4, dtype=torch.int64, device=dev)
    fst = torch.zeros(4, dtype=torch.float64, device=dev)

    def call():
        _lib.check(L.marl_reward_events_noalloc(vp(r.data_ptr()), E, A, vp(tot.data_ptr()), vp(fst.data_ptr()),
                                                vp(torch.cuda.current_stream(dev).cuda_stream)), "marl_reward_events_noalloc")

    call()  # outside any capture: the one-time address lookup happens here
    torch.cuda.synchronize()
    once_t, once_f = tot.clone(), fst.clone()
    reserved_before = torch.cuda.memory_reserved(dev)
    calls, steady = 1, False
    for _ in range(3):  # the device-wide figure can move for reasons outside this process: one quiet round suffices
        free_before, _ = torch.cuda.mem_get_info(dev)
        for _ in range(50):
            call()
        calls += 50
        torch.cuda.synchronize()
        steady = steady or torch.cuda.mem_get_info(dev)[0] == free_before
        if steady:
            break
    assert steady, "the device's free memory moved across no-allocation calls"
    assert torch.cuda.memory_reserved(dev) == reserved_before
    assert torch.equal(tot, once_t * calls)

    tot.zero_(); fst.zero_()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call()  # any allocation or other forbidden call in here would invalidate the capture
    assert int(tot.sum()) == 0  # capture records, it does not run
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(tot, once_t * 3)
    torch.testing.assert_close(fst, once_f * 3, rtol=1e-12, atol=0)
