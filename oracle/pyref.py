"""Per-env Python objects + the reference's serial loops, restated -- TEST INFRASTRUCTURE ONLY.

This is the shape of the reference's CPU path: one Python object per env behind the gym.Env
protocol (gymnasium 0.29.1 CartPoleEnv / AcrobotEnv under TimeLimit(500) and
RecordEpisodeStatistics(100); reference call sites src/env_cartpole.py:7-8,
src/env_acrobot.py:7-8), stepped one by one by SyncVecEnv.vec_step / rollout
(src/ppo.py:283-417) and followed by the T-iteration torch-CPU GAE loop (src/ppo.py:468-486).

Used (a) to cross-check ppo_oracle.c bit for bit (both use the host libm), and (b) as the
single-core "port" CPU baseline timed by bench.py.  PARITY UNPINNED at the physics level
(gymnasium is not installed; see ppo_oracle.c header).
"""
import math
from collections import deque

import numpy as np
import torch


class CartPoleEnv:
    gravity, masscart, masspole = 9.8, 1.0, 0.1
    total_mass = masspole + masscart
    length = 0.5
    polemass_length = masspole * length
    force_mag, tau = 10.0, 0.02
    theta_threshold_radians = 12 * 2 * math.pi / 360
    x_threshold = 2.4
    max_episode_steps = 500

    def __init__(self, rng=None):
        self.rng = rng if rng is not None else np.random.default_rng()
        self.state = None
        self.elapsed = 0

    def reset(self, u=None):
        if u is None:
            u = self.rng.random(4)
        self.state = tuple(float(-0.05 + (0.05 - -0.05) * x) for x in u)
        self.elapsed = 0
        return np.array(self.state, dtype=np.float32), {}

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if action == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + self.polemass_length * theta_dot**2 * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - self.masspole * costheta**2 / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = (x, x_dot, theta, theta_dot)
        terminated = bool(x < -self.x_threshold or x > self.x_threshold
                          or theta < -self.theta_threshold_radians or theta > self.theta_threshold_radians)
        self.elapsed += 1
        truncated = self.elapsed >= self.max_episode_steps
        return np.array(self.state, dtype=np.float32), 1.0, terminated, truncated, {}


class AcrobotEnv:
    dt = 0.2
    max_episode_steps = 500

    def __init__(self, rng=None):
        self.rng = rng if rng is not None else np.random.default_rng()
        self.state = None
        self.elapsed = 0

    def _ob(self):
        s = self.state
        if s.dtype == np.float32:  # right after reset
            return np.array([np.cos(s[0]), np.sin(s[0]), np.cos(s[1]), np.sin(s[1]), s[2], s[3]], dtype=np.float32)
        return np.array([math.cos(s[0]), math.sin(s[0]), math.cos(s[1]), math.sin(s[1]), s[2], s[3]],
                        dtype=np.float32)

    def reset(self, u=None):
        if u is None:
            u = self.rng.random(4)
        self.state = np.array([-0.1 + (0.1 - -0.1) * x for x in u]).astype(np.float32)
        self.elapsed = 0
        return self._ob(), {}

    @staticmethod
    def _dsdt(s):
        m1 = m2 = l1 = 1.0
        lc1 = lc2 = 0.5
        I1 = I2 = 1.0
        g = 9.8
        a = s[4]
        theta1, theta2, dtheta1, dtheta2 = s[0], s[1], s[2], s[3]
        cos, sin, pi = math.cos, math.sin, math.pi
        d1 = m1 * lc1**2 + m2 * (l1**2 + lc2**2 + 2 * l1 * lc2 * cos(theta2)) + I1 + I2
        d2 = m2 * (lc2**2 + l1 * lc2 * cos(theta2)) + I2
        phi2 = m2 * lc2 * g * cos(theta1 + theta2 - pi / 2.0)
        phi1 = (-m2 * l1 * lc2 * dtheta2**2 * sin(theta2)
                - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * sin(theta2)
                + (m1 * lc1 + m2 * l1) * g * cos(theta1 - pi / 2)
                + phi2)
        ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * dtheta1**2 * sin(theta2) - phi2) / (
            m2 * lc2**2 + I2 - d2**2 / d1)
        ddtheta1 = -(d2 * ddtheta2 + phi1) / d1
        return (dtheta1, dtheta2, ddtheta1, ddtheta2, 0.0)

    def step(self, a):
        s = [float(x) for x in self.state]
        y0 = s + [float(a - 1)]
        dt = self.dt - 0
        dt2 = dt / 2.0
        f = self._dsdt
        k1 = f(y0)
        k2 = f([y0[i] + dt2 * k1[i] for i in range(5)])
        k3 = f([y0[i] + dt2 * k2[i] for i in range(5)])
        k4 = f([y0[i] + dt * k3[i] for i in range(5)])
        ns = [y0[i] + dt / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) for i in range(4)]

        def wrap(x, m, M):
            diff = M - m
            while x > M:
                x = x - diff
            while x < m:
                x = x + diff
            return x

        pi = math.pi
        ns[0] = wrap(ns[0], -pi, pi)
        ns[1] = wrap(ns[1], -pi, pi)
        ns[2] = min(max(ns[2], -4 * pi), 4 * pi)
        ns[3] = min(max(ns[3], -9 * pi), 9 * pi)
        self.state = np.array(ns, dtype=np.float64)
        terminated = bool(-math.cos(ns[0]) - math.cos(ns[1] + ns[0]) > 1.0)
        reward = -1.0 if not terminated else 0.0
        self.elapsed += 1
        return self._ob(), reward, terminated, self.elapsed >= self.max_episode_steps, {}


class RecordEpisodeStatistics:
    def __init__(self, env, deque_size=100):
        self.env = env
        self.return_queue = deque(maxlen=deque_size)
        self.length_queue = deque(maxlen=deque_size)
        self.ep_ret, self.ep_len = np.float32(0.0), 0

    def reset(self, **kw):
        self.ep_ret, self.ep_len = np.float32(0.0), 0
        return self.env.reset(**kw)

    def step(self, action):
        obs, r, d, tr, info = self.env.step(action)
        self.ep_ret = np.float32(self.ep_ret + np.float32(r))
        self.ep_len += 1
        if d or tr:
            self.return_queue.append(self.ep_ret)
            self.length_queue.append(self.ep_len)
            self.ep_ret, self.ep_len = np.float32(0.0), 0
        return obs, r, d, tr, info


def make_env(kind, rng=None):
    env = CartPoleEnv(rng) if kind in (0, "cartpole") else AcrobotEnv(rng)
    return RecordEpisodeStatistics(env, 100)


def serial_rollout(envs, t_states, policy_fn, T):
    """The reference's rollout loop (src/ppo.py:320-398): per-env Python stepping, per-actor
    bookkeeping with per-element tensor writes.  policy_fn(states[N,obs]) -> (actions, logp, values).
    Returns the buffers, the ragged end_states and the carried t_states."""
    N = len(envs)
    od = t_states.shape[1]
    states = torch.zeros((T, N, od))
    actions = torch.zeros((T, N), dtype=torch.int32)
    rewards = torch.zeros((T, N))
    dones = torch.zeros((T, N), dtype=torch.int32)
    truncs = torch.zeros((T, N), dtype=torch.int32)
    values = torch.zeros((T, N))
    logps = torch.zeros((T, N))
    end_states = [[] for _ in range(N)]
    with torch.no_grad():
        for t in range(T):
            a, lp, v = policy_fn(t_states)
            a_np = a.cpu().numpy()
            new_states = torch.zeros((N, od))
            r_t = torch.zeros(N)
            d_t = torch.zeros(N, dtype=torch.int32)
            tr_t = torch.zeros(N, dtype=torch.int32)
            for i, env in enumerate(envs):  # :294-300
                s, r, d, tr, _ = env.step(np.squeeze(a_np[i]))
                new_states[i] = torch.tensor(s, dtype=torch.float32)
                r_t[i] = torch.tensor(r, dtype=torch.float32)
                d_t[i] = torch.tensor(d, dtype=torch.int32)
                tr_t[i] = torch.tensor(tr, dtype=torch.int32)
            for i in range(N):  # :342-388
                ended = d_t[i] + tr_t[i]
                can_reset = True
                if ended == 0 and t == T - 1:
                    ended += 1
                    tr_t[i] += 1
                    can_reset = False
                if ended > 0:
                    end_states[i].append(new_states[i].clone())
                    if can_reset:
                        new_states[i] = torch.tensor(envs[i].reset()[0], dtype=torch.float32)
            states[t], actions[t], rewards[t], dones[t], truncs[t] = t_states, a, r_t, d_t, tr_t
            values[t], logps[t] = v, lp
            t_states = new_states
    end = torch.cat([torch.stack(e) for e in end_states], dim=0)
    return dict(states=states, actions=actions, rewards=rewards, done=dones, trunc=truncs, values=values,
                log_probs=logps, end_states=end, t_states=t_states)


def gae_torch(rewards, values, end_values, done_flags, trunc_flags, gamma, lam):
    """The reference's T-iteration torch loop (src/ppo.py:463-488), restated."""
    T, N = rewards.shape
    adv = torch.zeros_like(rewards)
    ep_counts = torch.sum(done_flags + trunc_flags - done_flags * trunc_flags, dim=0, dtype=torch.int32)
    idx = torch.cumsum(ep_counts, dim=0) - 1
    nv = torch.zeros(N)
    na = torch.zeros(N)
    for t in reversed(range(T)):
        term = done_flags[t] + trunc_flags[t]
        nv = (1 - term) * nv + trunc_flags[t] * end_values[idx]
        na = (1 - term) * na
        idx = idx - term
        delta = rewards[t] + gamma * nv - values[t]
        adv[t] = delta + gamma * lam * na
        nv = values[t]
        na = adv[t]
    return adv


def losses_torch(prob_ratios, curr_values, prev_values, advantages, clip_ratio, norm_adv=1, clip_va_loss=1):
    """The reference's compute_losses in torch ops (src/ppo.py:498-525), restated; differentiable
    w.r.t. prob_ratios / curr_values like the original.  Pinned against the reference-run goldens in
    tests/test_oracle.py."""
    returns = advantages + prev_values                                             # :498
    if norm_adv:
        advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)  # :500-501
    clipped = torch.clamp(prob_ratios, 1 - clip_ratio, 1 + clip_ratio)             # :503
    policy_loss = -torch.minimum(prob_ratios * advantages, clipped * advantages).mean()   # :504-508
    if clip_va_loss:                                                               # :512-516
        sq = (returns - curr_values) ** 2
        vc = torch.clamp(curr_values, prev_values - clip_ratio, prev_values + clip_ratio)
        value_loss = 0.5 * torch.maximum(sq, (returns - vc) ** 2).mean()
    else:
        value_loss = 0.5 * ((returns - curr_values) ** 2).mean()                   # :518
    with torch.no_grad():                                                          # :520-525
        clip_frac = ((prob_ratios - 1.0).abs() > clip_ratio).float().mean()
        kl_div = (prob_ratios - 1 - torch.log(prob_ratios)).mean()
    return policy_loss, value_loss, clip_frac, kl_div
