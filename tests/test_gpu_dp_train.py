"""GPU: the data-parallel trainer executed with world_size 2 and compared with the single-process
run it must equal ("nothing to match except the single-GPU result"; the update loop
being reproduced is reference src/ppo.py:573-635).

Two fresh child processes (tests/dp_worker.py) share the one GPU over the gloo backend (RCCL wants a
GPU per rank; the collectives' semantics are the same), a third runs world_size 1 with the same seed:
  * parameters are bit-identical across the ranks after EVERY optimiser step;
  * the ranks' rollout buffers of update 0, concatenated along the env axis, equal the world-1
    buffers bit for bit (Philox streams keyed by global env id; broadcast initial parameters);
  * the first minibatch: the mean of the two ranks' losses equals the losses of the union minibatch
    computed here with the world-1 parameters and buffers, to 1e-5 (global advantage normalisation);
  * the number of optimiser steps per epoch and the global minibatch size equal the world-1 run's.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")
T, UPDATES = 32, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(out, world, kind, N, BATCH, joint=1, early_stop_reward=None, updates=UPDATES, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        if world > 1:
            env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), PPO_AMD_DIST_BACKEND="gloo")
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        env["PPO_AMD_DP_WATCHDOG"] = "200"               # a rank that hangs dumps every thread's stack and exits
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, "-u", WORKER, out, str(N), str(T), str(BATCH), str(updates), kind, str(joint),
                                       "none" if early_stop_reward is None else str(early_stop_reward)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(f"{out}.rank{r}", weights_only=True) for r in range(world)]


@pytest.mark.parametrize("kind,N,BATCH", [("cartpole", 256, 2048), ("acrobot", 2048, 16384)],
                         ids=["cartpole_256", "acrobot_2048_sharded_like_configs2"])
def test_world2_training_equals_the_single_process_run(tmp_path, kind, N, BATCH):
    from ppo_amd import Agent, ops
    from ppo_amd.envs import DeviceEnv
    r0, r1 = _launch(str(tmp_path / "dp2"), 2, kind, N, BATCH)
    (w1,) = _launch(str(tmp_path / "dp1"), 1, kind, N, BATCH)
    assert r0["world"] == 2 and r1["rank"] == 1 and w1["world"] == 1

    # broadcast: both ranks start from rank 0's parameters, which are the world-1 run's (same seed)
    for k in r0["init"]:
        assert torch.equal(r0["init"][k], r1["init"][k]) and torch.equal(r0["init"][k], w1["init"][k]), k
    # parameters after every optimiser step: bit-identical across ranks
    assert len(r0["steps"]) == len(r1["steps"]) > 0
    for a, b in zip(r0["steps"], r1["steps"]):
        assert (a["update"], a["epoch"], a["start"]) == (b["update"], b["epoch"], b["start"])
        assert torch.equal(a["params"], b["params"])
        assert a["kl_div"] == a["kl_div"]                       # finite
    # update 0's rollout: the shards side by side are the single-process rollout, bit for bit
    for k, whole in w1["buffers"].items():
        both = torch.cat([r0["buffers"][k], r1["buffers"][k]], dim=1)
        assert torch.equal(both, whole), k
    # optimiser steps per epoch / global minibatch size as in the single-process run
    per_epoch = T * N // BATCH
    first_epoch_2 = [s for s in r0["steps"] if s["update"] == 0 and s["epoch"] == 0]
    first_epoch_1 = [s for s in w1["steps"] if s["update"] == 0 and s["epoch"] == 0]
    assert len(first_epoch_2) == len(first_epoch_1) == per_epoch
    assert r0["steps"][0]["idx"].numel() + r1["steps"][0]["idx"].numel() == w1["steps"][0]["idx"].numel() == BATCH

    # first minibatch: mean of the ranks' losses == losses of the union minibatch with world-1 state
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=1, device="cuda")
    agent.load_state_dict(w1["init"])
    buf = {k: v.cuda() for k, v in w1["buffers"].items()}
    adv = ops.gae_dense(buf["rewards"], buf["values"], buf["ev_dense"], buf["done_flags"], buf["trunc_flags"], 0.99, 0.95)
    half = N // 2
    union = []
    for r, rec in enumerate((r0, r1)):
        li = rec["steps"][0]["idx"].long()
        t, n = li // half, li % half
        union.append(t * N + r * half + n)                       # flat index into the world-1 [T, N] buffers
    union = torch.cat(union).cuda()
    flat = lambda x: x.flatten(0, 1)
    with torch.no_grad():
        logits, values = agent.policy_and_values(flat(buf["states"])[union])
        pl, vl, cf, kl, ent = ops.ppo_losses_discrete(logits, flat(buf["actions"])[union], values,
                                                      flat(buf["log_probs"])[union], flat(buf["values"])[union],
                                                      flat(adv)[union], 0.2, 1, 1)
    for name, want in (("policy_loss", pl), ("critic_loss", vl), ("entropy", ent), ("kl_div", kl)):
        got = 0.5 * (r0["steps"][0][name] + r1["steps"][0][name])
        assert abs(got - float(want)) <= 1e-5 * abs(float(want)) + 2e-7, (name, got, float(want))
    # the job-level rate uses the slowest rank's time: both ranks report the same figure
    for a, b in zip(r0["metrics"], r1["metrics"]):
        assert a["utils/env_steps_per_sec"] == b["utils/env_steps_per_sec"] > 0
    # episode statistics and the rollout's mean reward are JOB-WIDE (reference src/ppo.py:404-417, :318 over all
    # envs): identical on both ranks in every update, and in update 0 -- the same rollout bit for bit -- equal to the
    # single-process run's
    same = lambda x, y: x == y or (x != x and y != y)
    stat_keys = ("metrics/ep_return_max", "utils/ep_return_0.05", "utils/ep_return_0.50", "utils/ep_return_0.95",
                 "metrics/ep_length_0.50", "metrics/roll_return")
    for a, b in zip(r0["metrics"], r1["metrics"]):
        for k in stat_keys:
            assert same(a[k], b[k]), (k, a[k], b[k])
    for k in stat_keys:
        assert same(r0["metrics"][0][k], w1["metrics"][0][k]), (k, r0["metrics"][0][k], w1["metrics"][0][k])
    if kind == "cartpole":                                      # (no Acrobot episode ends within 3 x 32 steps)
        assert r0["metrics"][0]["utils/ep_return_0.50"] == r0["metrics"][0]["utils/ep_return_0.50"]   # not NaN


def test_world2_split_networks_with_the_fused_update(tmp_path):
    """Split networks, two ranks, 4 096 envs: the fused minibatch step issues two gradient all-reduces per
    minibatch (policy half, value half) with little host work in between.  (Over gloo on one shared GPU this
    is the case that deadlocked inside all_reduce until dist._all_reduce drained the stream first.)  Both
    ranks finish, hold bit-identical parameters after every optimiser step, and start from the world-1
    run's rollout."""
    N, BATCH = 4096, 4096 * T // 4
    r0, r1 = _launch(str(tmp_path / "dps2"), 2, "cartpole", N, BATCH, joint=0)
    (w1,) = _launch(str(tmp_path / "dps1"), 1, "cartpole", N, BATCH, joint=0)
    assert len(r0["steps"]) == len(r1["steps"]) == len(w1["steps"]) > 0
    for a, b in zip(r0["steps"], r1["steps"]):
        assert torch.equal(a["params"], b["params"])
    for k, whole in w1["buffers"].items():
        assert torch.equal(torch.cat([r0["buffers"][k], r1["buffers"][k]], dim=1), whole), k
    # first minibatch: the ranks' mean losses are those of the world-1 run's first minibatch only in distribution
    # (different samples); the parameters after it must be close: same data, same update rule, 2 x 1/2 the samples
    d = (r0["steps"][-1]["params"] - w1["steps"][-1]["params"]).abs()
    assert float(d.max()) < 0.05


def test_world2_early_stop_reward_ends_both_ranks_together(tmp_path):
    """reference src/ppo.py:758 leaves the training loop when max_ep_return -- over ALL envs -- reaches
    early_stop_reward.  Under data parallelism that decision must be the same on every rank (a rank that left alone
    would leave the others in the next update's first collective for ever): the statistics are job-wide and the
    decision itself is a MAX all-reduce.  Few envs per rank, so that the ranks' own shards do reach the threshold
    at different updates; PPO_AMD_DIST_DEBUG checks that both ranks enter the same collectives in the same order."""
    N, BATCH, UPD = 16, 16 * T // 2, 40
    r0, r1 = _launch(str(tmp_path / "es2"), 2, "cartpole", N, BATCH, early_stop_reward=60.0, updates=UPD,
                     extra_env={"PPO_AMD_DIST_DEBUG": "1"})
    (w1,) = _launch(str(tmp_path / "es1"), 1, "cartpole", N, BATCH, early_stop_reward=60.0, updates=UPD)
    assert r0["updates_done"] == r1["updates_done"] < UPD               # stopped early, at the same update
    assert r0["collectives"] == r1["collectives"] > 0                    # and after the same number of collectives
    # job-wide statistics: both ranks log the same numbers, and at the stopping update the threshold is reached
    for a, b in zip(r0["metrics"], r1["metrics"]):
        for k in ("metrics/ep_return_max", "utils/ep_return_0.50", "utils/ep_return_0.05", "utils/ep_return_0.95",
                  "metrics/ep_length_0.50"):
            assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), (k, a[k], b[k])
    assert r0["metrics"][-1]["metrics/ep_return_max"] >= 60.0
    assert all(m["metrics/ep_return_max"] < 60.0 for m in r0["metrics"][:-1])
    # the job-wide running max is the max of the ranks' own
    for m, x, y in zip(r0["metrics"], r0["local_max"], r1["local_max"]):
        own = [v for v in (x, y) if v == v]
        if own:
            assert m["metrics/ep_return_max"] == max(own)
    # same envs, same seeds, same rollouts: the world-1 run sees the same episodes in its first rollout
    assert w1["metrics"][0]["metrics/ep_return_max"] == r0["metrics"][0]["metrics/ep_return_max"]
