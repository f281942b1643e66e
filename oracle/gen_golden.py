"""Generate tests/golden/ref_*.npz by RUNNING THE REFERENCE's own functions.

Run in the build container only (needs /root/reference):  python oracle/gen_golden.py

The reference's src/ppo.py imports `gymnasium` and `wandb` at module top (src/ppo.py:7-8);
neither is installed here and neither is used by the functions exercised below
(PPO.__init__, PPO.compute_advantages, PPO.compute_losses, SyncVecEnv.rollout with a
scripted fake env).  Two empty placeholder modules are registered so that the import
statement succeeds; no reference code is copied, and the fixtures hold data only (inputs
and the reference's outputs).  sys.dont_write_bytecode keeps /root/reference untouched.

TEST INFRASTRUCTURE ONLY.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Space:
        pass

    class Discrete(Space):
        def __init__(self, n):
            self.n, self.shape = n, ()

    class Box(Space):
        def __init__(self, low, high, shape):
            self.low, self.high, self.shape = low, high, tuple(shape)

    spaces.Space, spaces.Discrete, spaces.Box = Space, Discrete, Box
    gym.spaces, gym.Env, gym.Wrapper = spaces, object, object
    wandb = types.ModuleType("wandb")
    wandb.run = None
    sys.modules.setdefault("gymnasium", gym)
    sys.modules.setdefault("gymnasium.spaces", spaces)
    sys.modules.setdefault("wandb", wandb)
    sys.path.insert(0, REF)
    import ppo  # noqa: E402  (the reference module)
    return ppo, spaces


def rand_flags(rng, T, N, p_done, p_trunc, both=False, force_last=True):
    done = (rng.random((T, N)) < p_done).astype(np.int32)
    trunc = ((rng.random((T, N)) < p_trunc) & (done == 0)).astype(np.int32)
    if both:  # a few both-flags-set cells (src/ppo.py:476 gives terminations == 2)
        k = max(1, (T * N) // 200)
        tt, nn = rng.integers(0, T - 1, k), rng.integers(0, N, k)
        done[tt, nn] = 1
        trunc[tt, nn] = 1
    if force_last:  # rollout always ends a row T-1 with a flag (src/ppo.py:352-356)
        trunc[T - 1] = 1 - done[T - 1]
    return done, trunc


def gen_gae(ppo_mod):
    cases = {}
    rng = np.random.default_rng(20241218)
    shapes = [(8, 2), (20, 1), (128, 64), (125, 4), (256, 33), (1, 5), (3, 1)]
    gls = [(0.99, 0.95), (1.0, 1.0), (0.9, 0.0)]
    idx = 0
    for (T, N) in shapes:
        for (g, l) in gls:
            for mode in ("plain", "both", "nolast"):
                if mode == "both" and T < 4:
                    continue
                done, trunc = rand_flags(rng, T, N, 1 / 12, 1 / 30, both=(mode == "both"),
                                         force_last=(mode != "nolast"))
                r = rng.standard_normal((T, N)).astype(np.float32)
                v = rng.standard_normal((T, N)).astype(np.float32)
                # reference rollout appends ONE end state per (t,n) with any flag (:377-379)
                E = int(((done + trunc) > 0).sum())
                if mode == "nolast" and E == 0:
                    E = 1  # end_values must be indexable (negative index wraps onto it)
                ev = rng.standard_normal((max(E, 1),)).astype(np.float32)
                p = ppo_mod.PPO(g, l, 1, 0, 0, 1, False)
                try:
                    adv = p.compute_advantages(torch.tensor(r), torch.tensor(v), torch.tensor(ev),
                                               torch.tensor(done), torch.tensor(trunc)).numpy()
                    err = 0
                except IndexError:
                    adv = np.zeros_like(r)
                    err = 1
                pre = f"c{idx}_"
                cases.update({pre + "r": r, pre + "v": v, pre + "ev": ev, pre + "done": done,
                              pre + "trunc": trunc, pre + "gl": np.array([g, l]), pre + "adv": adv,
                              pre + "err": np.array(err)})
                idx += 1
    cases["n_cases"] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, "ref_gae.npz"), **cases)
    print("gae cases:", idx)


def gen_test_tables(ppo_mod):
    """The reference's own known-answer inputs (tests/test_ppo.py:25-29, :52-56, :101-105)
    re-run through the reference at full precision (the test file prints 4 decimals)."""
    p = ppo_mod.PPO(0.99, 0.95, 1, 0, 0, 1, False)
    r1 = np.array([[1.0, 1.0], [1.0, 0.5], [1.0, 2.0], [1.0, 1.0], [1.0, 0.0], [1.0, 1.0], [1.0, 0.5], [1.0, 2.0]], np.float32)
    v1 = np.array([[2.0, 0.6], [2.2, 1.2], [1.8, 1.0], [1.8, 1.0], [1.5, 0.8], [2.2, 1.0], [2.4, 0.6], [1.4, 1.4]], np.float32)
    e1 = np.array([1.6, 1.8, 0.6], np.float32)
    d1 = np.array([[0, 0], [0, 0], [0, 0], [1, 0], [0, 1], [0, 0], [0, 0], [0, 0]], np.int32)
    t1 = np.array([[0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 1]], np.int32)
    a1 = p.compute_advantages(*(torch.tensor(x) for x in (r1, v1, e1, d1, t1))).numpy()
    r2 = np.array([1.0, 1.0, 2.0, 1.0, 1.0, 2.0, 0.0, 1.0, 2.0, 0.0, 1.0, 1.0, 1.0, 3.0, 0.0, 1.0, 1.0, 3.0, 2.0, 1.0], np.float32)[:, None]
    v2 = np.array([2.0, 1.8, 1.6, 2.0, 1.8, 1.8, 1.8, 2.2, 2.0, 2.6, 1.6, 1.6, 2.4, 2.4, 2.4, 2.0, 1.8, 2.0, 2.0, 2.4], np.float32)[:, None]
    e2 = np.array([2.4, 1.2, 1.8, 1.4, 2.2], np.float32)
    d2 = np.array([0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0], np.int32)[:, None]
    t2 = np.array([0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1], np.int32)[:, None]
    a2 = p.compute_advantages(*(torch.tensor(x) for x in (r2, v2, e2, d2, t2))).numpy()
    # both-flags probe recorded in SURVEY.md 8c
    pb = p.compute_advantages(torch.ones(4, 1), torch.tensor([[1.0], [2.0], [3.0], [4.0]]), torch.tensor([5.0, 6.0]),
                              torch.tensor([[0], [1], [0], [0]], dtype=torch.int32),
                              torch.tensor([[0], [1], [0], [1]], dtype=torch.int32)).numpy()
    ratios = np.array([1.21, 0.99, 1.01, 1.05, 0.75], np.float32)
    cv = np.array([2.0, 1.8, 2.4, 0.8, 2.4], np.float32)
    pv = np.array([1.8, 1.8, 2.0, 1.0, 2.2], np.float32)
    ad = np.array([1.9, 1.9, 2.3, 0.8, 2.3], np.float32)
    l0 = [x.item() for x in p.compute_losses(*(torch.tensor(x) for x in (ratios, cv, pv, ad)), 0.2)]
    p1 = ppo_mod.PPO(0.99, 0.95, 1, 1, 0, 1, False)
    l1 = [x.item() for x in p1.compute_losses(*(torch.tensor(x) for x in (ratios, cv, pv, ad)), 0.2)]
    np.savez_compressed(os.path.join(OUT, "ref_test_tables.npz"), r1=r1, v1=v1, e1=e1, d1=d1, t1=t1, a1=a1,
                        r2=r2, v2=v2, e2=e2, d2=d2, t2=t2, a2=a2, probe_both=pb, ratios=ratios, cv=cv, pv=pv,
                        ad=ad, loss_clipva0=np.array(l0, np.float32), loss_clipva1=np.array(l1, np.float32))
    print("tables: a1[0]=", a1[0], "loss0=", l0, "loss1=", l1, "probe=", pb.ravel())


def gen_loss(ppo_mod):
    cases = {}
    rng = np.random.default_rng(7)
    idx = 0
    for B in (1, 2, 5, 128, 1000, 4097):
        for norm_adv in (0, 1):
            for clip_va in (0, 1):
                for clip in (0.2, 0.1):
                    if B == 1 and norm_adv:
                        continue  # std of one sample is NaN in torch; covered by a host test
                    old_lp = -np.abs(rng.standard_normal(B)).astype(np.float32)
                    new_lp = (old_lp + 0.15 * rng.standard_normal(B).astype(np.float32)).astype(np.float32)
                    old_v = rng.standard_normal(B).astype(np.float32)
                    new_v = (old_v + 0.3 * rng.standard_normal(B).astype(np.float32)).astype(np.float32)
                    adv = rng.standard_normal(B).astype(np.float32)
                    if B >= 5:  # exact ties / boundaries: rho == 1, adv == 0, v == v_old
                        new_lp[0] = old_lp[0]
                        adv[1] = 0.0
                        new_v[2] = old_v[2]
                    p = ppo_mod.PPO(0.99, 0.95, norm_adv, clip_va, 0, 1, False)
                    nl = torch.tensor(new_lp, requires_grad=True)
                    nv = torch.tensor(new_v, requires_grad=True)
                    ratio = torch.exp(nl - torch.tensor(old_lp))  # src/ppo.py:595
                    ratio.retain_grad()
                    pl, vl, cf, kl = p.compute_losses(ratio, nv, torch.tensor(old_v), torch.tensor(adv), clip)
                    pl.backward(retain_graph=True)
                    g_lp = nl.grad.clone().numpy()
                    g_ratio = ratio.grad.clone().numpy()
                    vl.backward()
                    g_v = nv.grad.clone().numpy()
                    pre = f"c{idx}_"
                    cases.update({pre + "new_lp": new_lp, pre + "old_lp": old_lp, pre + "new_v": new_v,
                                  pre + "old_v": old_v, pre + "adv": adv, pre + "ratio": ratio.detach().numpy(),
                                  pre + "cfg": np.array([norm_adv, clip_va, clip]),
                                  pre + "out4": np.array([pl.item(), vl.item(), cf.item(), kl.item()], np.float32),
                                  pre + "g_lp": g_lp, pre + "g_ratio": g_ratio, pre + "g_v": g_v})
                    idx += 1
    cases["n_cases"] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, "ref_loss.npz"), **cases)
    print("loss cases:", idx)


def gen_loss_logits(ppo_mod, spaces):
    """Discrete head on the update path: the reference's Agent.get_actions_and_values
    (src/ppo.py:166-216, Categorical log_prob + entropy) -> ratio (:595) -> compute_losses
    (:490-527), with autograd gradients w.r.t. the policy head's output (the logits)."""
    cases = {}
    rng = np.random.default_rng(11)
    idx = 0
    for A in (2, 3, 6):
        for B in (1, 5, 257, 2048):
            for (norm_adv, clip_va) in ((0, 0), (1, 1)):
                if B == 1 and norm_adv:
                    continue
                torch.manual_seed(100 + idx)
                agent = ppo_mod.Agent(spaces.Box(-1.0, 1.0, (4,)), spaces.Discrete(A), conv_net=0, joint_net=1,
                                      device="cpu")
                with torch.no_grad():
                    agent.policy.weight.mul_(150.0)  # gain-0.01 init gives near-uniform logits; spread them
                grabbed = {}

                def hook(_m, _inp, out):
                    out.retain_grad()
                    grabbed["logits"] = out

                h = agent.policy.register_forward_hook(hook)
                states = torch.tensor(rng.standard_normal((B, 4)).astype(np.float32))
                actions = torch.tensor(rng.integers(0, A, B).astype(np.int32))
                _, new_lp, new_v, ent = agent.get_actions_and_values(states, actions)
                h.remove()
                new_v.retain_grad()
                logits = grabbed["logits"]
                old_lp = (new_lp.detach().numpy() + 0.15 * rng.standard_normal(B)).astype(np.float32)
                old_v = (new_v.detach().numpy() + 0.3 * rng.standard_normal(B)).astype(np.float32)
                adv = rng.standard_normal(B).astype(np.float32)
                p = ppo_mod.PPO(0.99, 0.95, norm_adv, clip_va, 0, 1, False)
                ratio = torch.exp(new_lp - torch.tensor(old_lp))  # src/ppo.py:595
                pl, vl, cf, kl = p.compute_losses(ratio, new_v, torch.tensor(old_v), torch.tensor(adv), 0.2)
                pl.backward(retain_graph=True)
                g_pl = logits.grad.clone().numpy()
                logits.grad = None
                ent.backward(retain_graph=True)
                g_ent = logits.grad.clone().numpy()
                new_v.grad = None
                vl.backward()
                g_v = new_v.grad.clone().numpy()
                pre = f"c{idx}_"
                cases.update({pre + "logits": logits.detach().numpy(), pre + "actions": actions.numpy(),
                              pre + "old_lp": old_lp, pre + "new_v": new_v.detach().numpy(), pre + "old_v": old_v,
                              pre + "adv": adv, pre + "cfg": np.array([norm_adv, clip_va, 0.2]),
                              pre + "new_lp": new_lp.detach().numpy(),
                              pre + "out5": np.array([pl.item(), vl.item(), cf.item(), kl.item(), ent.item()],
                                                     np.float32),
                              pre + "g_pl": g_pl, pre + "g_ent": g_ent, pre + "g_v": g_v})
                idx += 1
    cases["n_cases"] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, "ref_loss_logits.npz"), **cases)
    print("loss-logits cases:", idx)


class ScriptedEnv:
    """Deterministic fake env (gym.Env protocol, src/ppo.py:272,296): obs encodes
    (env id, episode number, step in episode); done/trunc follow a fixed schedule."""

    def __init__(self, eid, ep_len, trunc_at):
        self.eid, self.ep_len, self.trunc_at = eid, ep_len, trunc_at
        self.episode, self.k = -1, 0

    def _obs(self):
        return np.array([self.eid, self.episode, self.k, 0.5], np.float32)

    def reset(self):
        self.episode += 1
        self.k = 0
        return self._obs(), {}

    def step(self, action):
        self.k += 1
        done = (self.k == self.ep_len[self.episode % len(self.ep_len)])
        trunc = (self.k == self.trunc_at) and not done
        return self._obs(), float(self.eid + 1 + 0.25 * self.k), done, trunc, {}


def gen_rollout(ppo_mod, spaces):
    """Golden for the rollout control logic (src/ppo.py:304-402): forced last-step
    truncation without reset, terminal-obs capture, auto-reset, actor-major end_states,
    carry-over of t_states into a second rollout."""
    torch.manual_seed(0)
    N, T = 3, 12
    scheds = [([5, 3], 99), ([20], 7), ([4, 100], 99)]
    it = iter(range(N))

    def env_fn():
        i = next(it)
        return ScriptedEnv(i, *scheds[i])

    box = spaces.Box(-1.0, 1.0, (4,))
    disc = spaces.Discrete(2)
    agent = ppo_mod.Agent(box, disc, conv_net=0, joint_net=1, device="cpu")
    os.makedirs("/tmp/ppo_golden_renders", exist_ok=True)
    cwd = os.getcwd()
    os.chdir("/tmp/ppo_golden_renders")  # SyncVecEnv.__init__ makes ./renders/misc (:246)
    try:
        ve = ppo_mod.SyncVecEnv(env_fn, N, T, 0, 0.0, agent)
        out = {}
        for k in range(2):
            ve.rollout()
            pre = f"r{k}_"
            out.update({pre + "states": ve.states.detach().numpy(), pre + "rewards": ve.rewards.numpy(),
                        pre + "done": ve.done_flags.numpy(), pre + "trunc": ve.trunc_flags.numpy(),
                        pre + "end_states": ve.end_states.numpy(), pre + "t_states": ve.t_states.numpy(),
                        pre + "total_return": np.array(ve.total_return),
                        pre + "global_steps": np.array(ve.global_steps)})
    finally:
        os.chdir(cwd)
    out["sched_len0"] = np.array(scheds[0][0]); out["sched_len1"] = np.array(scheds[1][0])
    out["sched_len2"] = np.array(scheds[2][0]); out["trunc_at"] = np.array([s[1] for s in scheds])
    np.savez_compressed(os.path.join(OUT, "ref_rollout.npz"), **out)
    print("rollout: end_states", out["r0_end_states"].shape, out["r1_end_states"].shape)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    ppo_mod, spaces = import_reference()
    gen_test_tables(ppo_mod)
    gen_gae(ppo_mod)
    gen_loss(ppo_mod)
    gen_rollout(ppo_mod, spaces)
    gen_loss_logits(ppo_mod, spaces)
