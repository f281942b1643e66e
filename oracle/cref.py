"""ctypes binding of oracle/ppo_oracle.c -- TEST INFRASTRUCTURE ONLY (see package docstring).

All functions take and return numpy arrays; nothing here touches a GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libppo_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ppo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def gae_ragged(rewards, values, end_values, done, trunc, gamma, lam):
    """src/ppo.py:452-488.  Raises IndexError where torch would."""
    r, v, ev, d, tr = _f32(rewards), _f32(values), _f32(end_values), _i32(done), _i32(trunc)
    T, N = r.shape
    adv = np.empty((T, N), np.float32)
    scratch = np.empty((N,), np.int64)
    rc = lib().oracle_gae_ragged(
        _p(r, C.c_float), _p(v, C.c_float), _p(ev, C.c_float), C.c_long(ev.size),
        _p(d, C.c_int), _p(tr, C.c_int), C.c_double(gamma), C.c_double(lam),
        _p(adv, C.c_float), C.c_int(T), C.c_int(N), _p(scratch, C.c_long))
    if rc != 0:
        raise IndexError("end_values index out of range (as torch would raise)")
    return adv


def gae_dense(rewards, values, end_value_dense, done, trunc, gamma, lam, want_returns=False):
    r, v, ev, d, tr = _f32(rewards), _f32(values), _f32(end_value_dense), _i32(done), _i32(trunc)
    T, N = r.shape
    adv = np.empty((T, N), np.float32)
    ret = np.empty((T, N), np.float32) if want_returns else None
    lib().oracle_gae_dense(
        _p(r, C.c_float), _p(v, C.c_float), _p(ev, C.c_float), _p(d, C.c_int), _p(tr, C.c_int),
        C.c_double(gamma), C.c_double(lam), _p(adv, C.c_float), _p(ret, C.c_float),
        C.c_int(T), C.c_int(N))
    return (adv, ret) if want_returns else adv


def ragged_to_dense(end_values, done, trunc):
    """Scatter the actor-major ragged end_values list (src/ppo.py:401-402) to [T,N]."""
    d, tr = _i32(done), _i32(trunc)
    T, N = d.shape
    flag = ((d + tr) > 0)
    dense = np.zeros((T, N), np.float32)
    n_idx, t_idx = np.nonzero(flag.T)  # actor-major, time-ascending
    dense[t_idx, n_idx] = _f32(end_values)[: n_idx.size]
    return dense


def adv_stats(adv):
    a = _f32(adv).ravel()
    out = np.empty((3,), np.float64)
    lib().oracle_adv_stats(_p(a, C.c_float), C.c_long(a.size), _p(out, C.c_double))
    return out


def loss(x, old_lp, new_v, old_v, adv, clip, norm_adv, clip_va, ratio_input, stats3=None):
    """src/ppo.py:490-527 (+:595).  Returns (out4, g_x, g_v)."""
    x, nv, ov, a = _f32(x).ravel(), _f32(new_v).ravel(), _f32(old_v).ravel(), _f32(adv).ravel()
    olp = None if old_lp is None else _f32(old_lp).ravel()
    B = x.size
    if stats3 is None:
        stats3 = adv_stats(a)
    stats3 = np.ascontiguousarray(stats3, np.float64)
    out4 = np.empty((4,), np.float32)
    gx = np.empty((B,), np.float32)
    gv = np.empty((B,), np.float32)
    lib().oracle_loss(
        _p(x, C.c_float), _p(olp, C.c_float), _p(nv, C.c_float), _p(ov, C.c_float),
        _p(a, C.c_float), _p(stats3, C.c_double), C.c_double(clip), C.c_int(int(norm_adv)),
        C.c_int(int(clip_va)), C.c_int(int(ratio_input)), C.c_long(B),
        _p(out4, C.c_float), _p(gx, C.c_float), _p(gv, C.c_float))
    return out4, gx, gv


def loss_logits(logits, actions, old_lp, new_v, old_v, adv, clip, norm_adv, clip_va, stats3=None):
    """Categorical head on raw logits (src/ppo.py:186-189,212-214) + :595 + :490-527.
    Returns (out5, g_logits_pl, g_logits_ent, g_v)."""
    lg = np.ascontiguousarray(logits, np.float32)
    B, A = lg.shape
    act = _i32(actions).ravel()
    olp, nv, ov, a = _f32(old_lp).ravel(), _f32(new_v).ravel(), _f32(old_v).ravel(), _f32(adv).ravel()
    if stats3 is None:
        stats3 = adv_stats(a)
    stats3 = np.ascontiguousarray(stats3, np.float64)
    out5 = np.empty((5,), np.float32)
    gpl, gent = np.empty((B, A), np.float32), np.empty((B, A), np.float32)
    gv, s1, s2 = np.empty((B,), np.float32), np.empty((B,), np.float32), np.empty((B,), np.float32)
    lib().oracle_loss_logits(
        _p(lg, C.c_float), C.c_int(A), _p(act, C.c_int), _p(olp, C.c_float), _p(nv, C.c_float),
        _p(ov, C.c_float), _p(a, C.c_float), _p(stats3, C.c_double), C.c_double(clip),
        C.c_int(int(norm_adv)), C.c_int(int(clip_va)), C.c_long(B), _p(out5, C.c_float),
        _p(gpl, C.c_float), _p(gent, C.c_float), _p(gv, C.c_float), _p(s1, C.c_float), _p(s2, C.c_float))
    return out5, gpl, gent, gv


def philox4x32_10(ctr, key):
    c = np.ascontiguousarray(ctr, np.uint32)
    k = np.ascontiguousarray(key, np.uint32)
    o = np.empty((4,), np.uint32)
    lib().oracle_philox4x32_10(_p(c, C.c_uint32), _p(k, C.c_uint32), _p(o, C.c_uint32))
    return o


def reset_uniform4(seed, env_id, episode):
    u = np.empty((4,), np.float64)
    lib().oracle_reset_uniform4(C.c_uint64(seed), C.c_uint64(env_id), C.c_uint32(episode), _p(u, C.c_double))
    return u


def sample_uniform(seed, env_base, step, N):
    """The [0,1) draws behind the sampled actions of policy step `step` for envs env_base..env_base+N-1."""
    u = np.empty((N,), np.float32)
    lib().oracle_sample_uniform_batch(C.c_uint64(seed), C.c_uint64(env_base), C.c_uint64(step), _p(u, C.c_float),
                                      C.c_int(N))
    return u


def div_by_total_mass_mismatches(n, seed=0, e_lo=-40, e_hi=20):
    """How many of n random doubles (exponent in [e_lo, e_hi)) the device's three-FMA x / 1.1 gets wrong."""
    f = lib().oracle_div_by_total_mass_mismatches
    f.restype = C.c_long
    return int(f(C.c_long(n), C.c_uint64(seed), C.c_int(e_lo), C.c_int(e_hi)))


def acrobot_terminal(state):
    """(flags int32 [n], lhs float64 [n]) of gymnasium's AcrobotEnv._terminal, -cos(s0) - cos(s1 + s0) > 1.0, for
    float64 states [n, 4] (glibc cos)."""
    st = np.ascontiguousarray(state, dtype=np.float64)
    n = st.shape[0]
    flags, lhs = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float64)
    f = lib().oracle_acrobot_terminal
    f.restype = C.c_int
    f(_p(st, C.c_double), C.c_long(n), _p(flags, C.c_int), _p(lhs, C.c_double))
    return flags, lhs


def shuffle_indices(n, seed, stream_id):
    """The permutation ppo_shuffle_indices makes (oracle_shuffle_indices), as int64 [n]."""
    out = np.empty(n, np.int64)
    lib().oracle_shuffle_indices(C.c_long(n), C.c_uint64(seed), C.c_uint64(stream_id), _p(out, C.c_int64))
    return out


def categorical_check(logits, u, actions, margin=1e-5):
    """Inverse-CDF sampling check in float64: returns (mismatches, ambiguous) -- the number of rows whose
    action differs from the first k with u * sum < cdf_k although u * sum is further than `margin`
    (relative to sum) from every CDF boundary, and the number of rows inside such a margin."""
    lg = np.asarray(logits, np.float64)
    e = np.exp(lg - lg.max(axis=1, keepdims=True))
    cdf = np.cumsum(e, axis=1)
    tot = cdf[:, -1]
    target = np.asarray(u, np.float64) * tot
    pick = np.minimum((target[:, None] >= cdf).sum(axis=1), lg.shape[1] - 1)
    near = (np.abs(cdf[:, :-1] - target[:, None]) <= margin * tot[:, None]).any(axis=1)
    bad = (pick != np.asarray(actions).reshape(-1)) & ~near
    return int(bad.sum()), int(near.sum())


KIND = {"cartpole": 0, "acrobot": 1}
OBS_DIM = {0: 4, 1: 6}


class BatchEnv:
    """N envs of one kind stepped serially on one core, with the rollout bookkeeping of
    src/ppo.py:342-398 folded in (see oracle_env_step_batch).  Mirrors the device-side
    state layout of ppo_amd.csrc so parity tests compare field by field."""

    def __init__(self, kind, N, seed=0, env_base=0, ring=100):
        self.kind = KIND[kind] if isinstance(kind, str) else int(kind)
        self.N, self.seed, self.env_base, self.ring = N, seed, env_base, ring
        self.od = OBS_DIM[self.kind]
        self.state = np.zeros((N, 4), np.float64)
        self.elapsed = np.zeros((N,), np.int32)
        self.ep_ret = np.zeros((N,), np.float32)
        self.ep_len = np.zeros((N,), np.int32)
        self.episode = np.zeros((N,), np.uint32)
        self.ret_ring = np.zeros((N, ring), np.float32)
        self.len_ring = np.zeros((N, ring), np.int32)
        self.ring_cnt = np.zeros((N,), np.int32)

    def reset(self, reset_draw=None):
        obs = np.empty((self.N, self.od), np.float32)
        rd = None if reset_draw is None else np.ascontiguousarray(reset_draw, np.float64)
        lib().oracle_env_reset_batch(
            C.c_int(self.kind), _p(self.state, C.c_double), _p(self.elapsed, C.c_int),
            _p(self.ep_ret, C.c_float), _p(self.ep_len, C.c_int), _p(self.episode, C.c_uint32),
            _p(rd, C.c_double), C.c_uint64(self.seed), C.c_uint64(self.env_base),
            _p(obs, C.c_float), C.c_int(self.N))
        return obs

    def step(self, action, is_last=False, reset_draw=None):
        a = _i32(action)
        N, od = self.N, self.od
        obs = np.empty((N, od), np.float32)
        term_obs = np.zeros((N, od), np.float32)
        reward = np.empty((N,), np.float32)
        done = np.empty((N,), np.int32)
        trunc = np.empty((N,), np.int32)
        rd = None if reset_draw is None else np.ascontiguousarray(reset_draw, np.float64)
        lib().oracle_env_step_batch(
            C.c_int(self.kind), _p(self.state, C.c_double), _p(self.elapsed, C.c_int),
            _p(self.ep_ret, C.c_float), _p(self.ep_len, C.c_int), _p(self.episode, C.c_uint32),
            _p(a, C.c_int), _p(rd, C.c_double), C.c_uint64(self.seed), C.c_uint64(self.env_base),
            _p(obs, C.c_float), _p(term_obs, C.c_float), _p(reward, C.c_float),
            _p(done, C.c_int), _p(trunc, C.c_int), _p(self.ret_ring, C.c_float),
            _p(self.len_ring, C.c_int), _p(self.ring_cnt, C.c_int), C.c_int(self.ring),
            C.c_int(N), C.c_int(int(is_last)))
        return obs, term_obs, reward, done, trunc
