"""GPU: ppo_episode_stats_incremental (C ABI) against np.percentile / np.max on the host's copy of the
deques (reference src/ppo.py:404-417) and, bit for bit, against the full-scan ppo_episode_stats, over
sequences of calls between which episodes are appended to the rings the way RecordEpisodeStatistics'
deque.append does (slot ring_cnt % ring, ring_cnt += 1).

Covered: nothing finished yet; calls with nothing new; more than `ring` new episodes of one env between
two calls; rings that have not wrapped; integer returns (CartPole), negative integers (Acrobot) -- the
paths that end without a full pass --; a drifting distribution (percentiles move to other key
prefixes: the level-1 histograms are rebuilt); non-integer floats (level 2 over the rings every call);
ring lengths other than 100; the state re-zeroed after the rings were rewritten.
ppo_episode_stats_incremental_ints (the one-launch direct-histogram form a rollout calls) runs beside it on a state
of its own: identical statistics for integer rings, NaN while a value that is not a small integer is in the window."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class SelState(C.Structure):      # mirror of csrc/stats.hip SelState, only to read the skip flags back
    _fields_ = [("n", C.c_uint64), ("rank", C.c_uint64 * 9), ("prefix", C.c_uint32 * 9), ("slot", C.c_int * 9),
                ("leader", C.c_int * 9), ("nlead_ret", C.c_int), ("nlead", C.c_int), ("lprefix", C.c_uint32 * 9),
                ("lslot", C.c_int * 9), ("ticket", C.c_uint32), ("low_bits", C.c_uint32), ("skip2", C.c_int),
                ("skip1", C.c_int), ("ntrack_ret", C.c_int), ("ntrack", C.c_int), ("tprefix", C.c_uint32 * 9),
                ("tslot", C.c_int * 9), ("nz", C.c_int64)]


def _ints_head(state):
    """(n, bad) of csrc/stats.hip IntsState, the head of a state block used by ppo_episode_stats_incremental_ints."""
    raw = state[:16].cpu().numpy().tobytes()
    return int.from_bytes(raw[:8], "little"), int.from_bytes(raw[8:16], "little", signed=True)


def _flags(state):
    raw = state[:C.sizeof(SelState)].cpu().numpy().tobytes()
    st = SelState.from_buffer_copy(raw)
    return st.skip1, st.skip2, st.nz, st.n


class HostRings:
    """The deques of N envs, kept the way the device rings are."""

    def __init__(self, N, ring):
        self.N, self.ring = N, ring
        self.ret = np.zeros((N, ring), np.float32)
        self.len = np.zeros((N, ring), np.int32)
        self.cnt = np.zeros(N, np.int32)

    def append(self, counts, draw_ret, draw_len):
        for n in np.nonzero(counts)[0]:
            k = int(counts[n])
            r, l = draw_ret(k), draw_len(k)
            for j in range(k):
                s = (self.cnt[n] + j) % self.ring
                self.ret[n, s], self.len[n, s] = r[j], l[j]
            self.cnt[n] += k

    def window(self):
        valid = np.arange(self.ring)[None, :] < np.minimum(self.cnt, self.ring)[:, None]
        return self.ret[valid], self.len[valid]


def _want(h):
    r, l = h.window()
    if r.size == 0:
        return None
    p = np.percentile(r.astype(np.float64), [5.0, 50.0, 95.0])
    return np.array([r.max(), p[0], p[1], p[2], np.percentile(l, 50.0)])


CASES = {
    # name: (ring, N, calls, returns(k, call), mean new episodes per env and call)
    "cartpole_ints": (100, 3000, 8, lambda rng, k, c: rng.integers(8, 60, k).astype(np.float32), 6),
    "acrobot_negative": (100, 2000, 6, lambda rng, k, c: -rng.integers(60, 500, k).astype(np.float32), 2),
    "drifting": (100, 1500, 10, lambda rng, k, c: rng.integers(8 + 12 * c, 40 + 30 * c, k).astype(np.float32), 30),
    "floats": (100, 1000, 5, lambda rng, k, c: rng.normal(100.0, 40.0, k).astype(np.float32), 5),
    "constant": (100, 500, 4, lambda rng, k, c: np.full(k, 9.0, np.float32), 3),
    "short_ring": (7, 2000, 6, lambda rng, k, c: rng.integers(1, 30, k).astype(np.float32), 3),
    "long_ring": (257, 300, 5, lambda rng, k, c: rng.integers(-50, 50, k).astype(np.float32), 90),
    "ring_one": (1, 700, 5, lambda rng, k, c: rng.integers(1, 30, k).astype(np.float32), 1),
}


@pytest.mark.parametrize("case", list(CASES))
def test_incremental_statistics_follow_the_deques(case):
    from ppo_amd import _lib
    lib = _lib.load()
    ring, N, calls, draw, mean_new = CASES[case]
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    h = HostRings(N, ring)
    d_ret, d_len, d_cnt = (torch.from_numpy(x).cuda() for x in (h.ret, h.len, h.cnt))
    nbytes = lib.ppo_episode_stats_state_bytes(N, ring)
    assert nbytes >= 8 * N * ring + 4 * N
    state = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(lib.ppo_episode_stats_workspace_bytes(), dtype=torch.uint8, device="cuda")
    s_inc = torch.full((5,), float("nan"), dtype=torch.float64, device="cuda")
    s_full = torch.full((5,), float("nan"), dtype=torch.float64, device="cuda")
    # the one-launch direct-histogram form for small integers (what a rollout calls), on a state block of its own
    state_i = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    s_int = torch.full((5,), float("nan"), dtype=torch.float64, device="cuda")
    running_max = None
    fast_calls = 0

    def call():
        assert lib.ppo_episode_stats_incremental(d_ret.data_ptr(), d_len.data_ptr(), d_cnt.data_ptr(), ring, N,
                                                 s_inc.data_ptr(), state.data_ptr(), state.numel(), None) == 0
        assert lib.ppo_episode_stats_incremental_ints(d_ret.data_ptr(), d_len.data_ptr(), d_cnt.data_ptr(), ring, N,
                                                      s_int.data_ptr(), state_i.data_ptr(), state_i.numel(), None) == 0
        assert lib.ppo_episode_stats(d_ret.data_ptr(), d_len.data_ptr(), d_cnt.data_ptr(), ring, N,
                                     s_full.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0
        torch.cuda.synchronize()
        return s_inc.cpu().numpy(), s_full.cpu().numpy()

    got, full = call()                                   # nothing has finished: untouched
    assert np.isnan(got).all() and np.isnan(full).all()
    assert _flags(state)[:2] == (1, 1)
    for c in range(calls):
        counts = rng.poisson(mean_new, N).astype(np.int32)
        counts[rng.integers(0, N, N // 10)] = 0          # envs with nothing new
        if c == 2:
            counts[:5] = 2 * ring + 3                    # more than a whole ring between two calls
        if c == 3:
            counts[:] = 0                                # a call with nothing new at all
        h.append(counts, lambda k: draw(rng, k, c), lambda k: rng.integers(1, 501, k).astype(np.int32))
        d_ret.copy_(torch.from_numpy(h.ret)); d_len.copy_(torch.from_numpy(h.len)); d_cnt.copy_(torch.from_numpy(h.cnt))
        got, full = call()
        want = _want(h)
        running_max = want[0] if running_max is None else max(running_max, want[0])
        assert np.array_equal(got, full), (case, c, got, full)           # bit for bit the full scan's result
        assert got[0] == running_max
        assert np.allclose(got[1:4], want[1:4], rtol=1e-12, atol=0.0) and got[4] == want[4], (case, c, got, want)
        skip1, skip2, nz, n = _flags(state)
        assert n == h.window()[0].size
        assert (nz == 0) == (case != "floats")
        assert skip2 == (1 if case != "floats" else 0)
        fast_calls += int(skip1 == 1 and skip2 == 1)
        ints = s_int.cpu().numpy()
        if case != "floats":                             # integer-valued: the same numbers without the third launch
            assert np.array_equal(ints, got), (case, c, ints, got)
        else:                                            # promise broken: NaN, never a wrong value
            assert _ints_head(state_i)[1] != 0 and np.isnan(ints).all(), (case, c, ints)
        assert _ints_head(state_i)[0] == (n if case != "floats" else _ints_head(state_i)[0])
    if case in ("cartpole_ints", "acrobot_negative", "constant"):
        assert fast_calls >= calls - 2, fast_calls       # steady distributions: no full pass after the first calls
    # the rings rewritten behind the state's back (a restored checkpoint): zero the state, start over
    h2 = HostRings(N, ring)
    h2.append(rng.integers(0, 2 * ring, N).astype(np.int32), lambda k: draw(rng, k, 1),
              lambda k: rng.integers(1, 501, k).astype(np.int32))
    d_ret.copy_(torch.from_numpy(h2.ret)); d_len.copy_(torch.from_numpy(h2.len)); d_cnt.copy_(torch.from_numpy(h2.cnt))
    state.zero_(); state_i.zero_()
    s_inc.fill_(float("nan")); s_full.fill_(float("nan")); s_int.fill_(float("nan"))
    got, full = call()
    assert np.array_equal(got, full)
    if case != "floats":
        assert np.array_equal(s_int.cpu().numpy(), got)
    want = _want(h2)
    assert got[0] == want[0] and np.allclose(got[1:4], want[1:4], rtol=1e-12, atol=0.0) and got[4] == want[4]


def test_incremental_statistics_at_bench_scale_through_rollouts():
    """N = 65536 CartPole envs stepped by the whole-rollout kernel: after every rollout the incremental
    statistics equal the full scan's bit for bit, and from the third rollout on no full pass runs."""
    from ppo_amd import Agent, SyncVecEnv, _lib
    from ppo_amd.envs import DeviceEnv
    lib = _lib.load()
    N, T = 65536, 128
    probe = DeviceEnv("cartpole")
    agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
    ve = SyncVecEnv(lambda: DeviceEnv("cartpole", seed=2), N, T, 0, 0.0, agent, seed=2)
    b = ve.batch
    state = torch.zeros(lib.ppo_episode_stats_state_bytes(N, b.ring), dtype=torch.uint8, device="cuda")
    ws = torch.zeros(lib.ppo_episode_stats_workspace_bytes(), dtype=torch.uint8, device="cuda")
    s_inc = torch.full((5,), float("nan"), dtype=torch.float64, device="cuda")
    s_full = torch.full((5,), float("nan"), dtype=torch.float64, device="cuda")
    fast = []
    for it in range(8):
        ve.rollout()
        args = (b.ret_ring.data_ptr(), b.len_ring.data_ptr(), b.ring_cnt.data_ptr(), b.ring, N)
        assert lib.ppo_episode_stats_incremental(*args, s_inc.data_ptr(), state.data_ptr(), state.numel(), None) == 0
        assert lib.ppo_episode_stats(*args, s_full.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(s_inc, s_full), (it, s_inc, s_full)
        skip1, skip2, nz, n = _flags(state)
        assert nz == 0 and n == int(b.ring_cnt.clamp(max=b.ring).sum())
        fast.append(skip1 == 1 and skip2 == 1)
    r, l = b.episode_stats()
    want = np.percentile(r.cpu().numpy().astype(np.float64), [5.0, 50.0, 95.0])
    assert np.allclose(s_inc.cpu().numpy()[1:4], want, rtol=1e-12, atol=0.0)
    assert all(fast[2:]), fast
    ve._flush_stats()                                    # the rollouts' own statistics: the one-launch integer form
    assert torch.equal(ve._stats5, s_inc), (ve._stats5, s_inc)


def test_direct_histogram_form_bounds_and_recovery():
    """ppo_episode_stats_incremental_ints at the edges of its promise: returns of exactly +-512 and lengths of 512 are
    counted; 513 (an integer, but outside the bins), 0.5 and a length of 513 make the statistics NaN for exactly as
    long as they are in the window, and the numbers are right again once the ring has pushed them out."""
    from ppo_amd import _lib
    lib = _lib.load()
    ring, N = 4, 96
    rng = np.random.default_rng(5)
    h = HostRings(N, ring)
    d_ret, d_len, d_cnt = (torch.from_numpy(x).cuda() for x in (h.ret, h.len, h.cnt))
    state = torch.zeros(lib.ppo_episode_stats_state_bytes(N, ring), dtype=torch.uint8, device="cuda")
    s5 = torch.full((5,), float("nan"), dtype=torch.float64, device="cuda")

    def push(counts, draw_ret, draw_len):
        h.append(counts, draw_ret, draw_len)
        d_ret.copy_(torch.from_numpy(h.ret)); d_len.copy_(torch.from_numpy(h.len)); d_cnt.copy_(torch.from_numpy(h.cnt))
        assert lib.ppo_episode_stats_incremental_ints(d_ret.data_ptr(), d_len.data_ptr(), d_cnt.data_ptr(), ring, N,
                                                      s5.data_ptr(), state.data_ptr(), state.numel(), None) == 0
        torch.cuda.synchronize()
        return s5.cpu().numpy()

    every = np.full(N, 3, np.int32)
    got = push(every, lambda k: rng.choice([-512.0, 512.0, 0.0, 7.0], k).astype(np.float32),
               lambda k: rng.choice([1, 512, 500], k).astype(np.int32))
    want = _want(h)
    assert got[0] == want[0] == 512.0 and np.allclose(got[1:4], want[1:4], rtol=1e-12, atol=0.0) and got[4] == want[4]
    assert _ints_head(state) == (3 * N, 0)
    one = np.zeros(N, np.int32); one[5] = 1
    for bad_ret, bad_len in ((513.0, 10), (0.5, 10), (3.0, 513), (float("inf"), 3)):
        got = push(one, lambda k: np.full(k, bad_ret, np.float32), lambda k: np.full(k, bad_len, np.int32))
        assert np.isnan(got).all() and _ints_head(state)[1] == 1, (bad_ret, bad_len, got)
        four = np.zeros(N, np.int32); four[5] = ring             # the ring of env 5 turns over: the value leaves the window
        got = push(four, lambda k: rng.integers(-20, 20, k).astype(np.float32), lambda k: rng.integers(1, 400, k).astype(np.int32))
        want = _want(h)
        assert _ints_head(state)[1] == 0
        # (the running maximum restarts after a NaN: it is the window's maximum now)
        assert got[0] == want[0] and np.allclose(got[1:4], want[1:4], rtol=1e-12, atol=0.0) and got[4] == want[4], (got, want)
