"""CPU suite: the data-parallel glue (ppo_amd/dist.py) under world_size-2 gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ppo_amd import dist as D
    from ppo_amd import Agent, spaces
    from ppo_amd.ops import merge_stats
    assert D.init_from_env("gloo")
    assert D.world_size() == world and D.rank() == rank
    torch.manual_seed(100 + rank)  # different init per rank -> broadcast must equalise
    agent = Agent(spaces.Box(-1.0, 1.0, (4,)), spaces.Discrete(2), 0, 1, "cpu")
    D.broadcast_params(agent)
    flat = torch.cat([p.data.reshape(-1) for p in agent.parameters()])
    # per-rank gradients, then one bucketed all-reduce (mean)
    x = torch.full((8, 4), float(rank + 1))
    _, lp, v, ent = agent.get_actions_and_values(x, torch.zeros(8, dtype=torch.int32))
    (lp.mean() + v.mean()).backward()
    local = torch.cat([p.grad.reshape(-1) for p in agent.parameters()]).clone()
    D.allreduce_grads(list(agent.parameters()))
    avg = torch.cat([p.grad.reshape(-1) for p in agent.parameters()])
    # advantage statistics: each rank holds half of a global minibatch
    g = torch.Generator().manual_seed(5)
    full = torch.randn(1000, generator=g, dtype=torch.float64) * 3 + 1
    mine = full[rank::world]
    stats = torch.stack([torch.tensor(float(mine.numel()), dtype=torch.float64), mine.mean(),
                         ((mine - mine.mean()) ** 2).sum()])
    merged = D.merged_adv_stats(stats)
    kl = D.mean_scalar(torch.tensor(float(rank)))
    # the same for a whole epoch's table [num_mb, 3] in one collective: row m must equal the per-minibatch merge
    table, singles = [], []
    for m in range(3):
        part = (full[m * 300:(m + 1) * 300 + 100])[rank::world] * (m + 1.0)
        st = torch.stack([torch.tensor(float(part.numel()), dtype=torch.float64), part.mean(), ((part - part.mean()) ** 2).sum()])
        table.append(st)
        singles.append(D.merged_adv_stats(st))
    epoch = D.merged_adv_stats_epoch(torch.stack(table))
    assert epoch.shape == (3, 3) and epoch.is_contiguous() and torch.equal(epoch, torch.stack(singles))
    out[rank] = dict(flat=flat, local=local, avg=avg, merged=merged, kl=kl,
                     want=torch.stack([torch.tensor(1000.0, dtype=torch.float64), full.mean(),
                                       ((full - full.mean()) ** 2).sum()]))
    td.barrier()
    td.destroy_process_group()


def test_dp_glue_world2():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
        assert torch.equal(r0["flat"], r1["flat"])                               # broadcast
        assert torch.allclose(r0["avg"], (r0["local"] + r1["local"]) / 2)        # mean all-reduce
        assert torch.equal(r0["avg"], r1["avg"])
        assert torch.allclose(r0["merged"], r0["want"], rtol=1e-12)              # Chan merge == global stats
        assert torch.equal(r0["merged"], r1["merged"])
        assert float(r0["kl"]) == 0.5 and float(r1["kl"]) == 0.5


def test_single_process_is_identity():
    from ppo_amd import dist as D
    assert D.world_size() == 1 and D.rank() == 0 and not D.is_dist()
    s = torch.tensor([3.0, 1.0, 2.0], dtype=torch.float64)
    assert D.merged_adv_stats(s) is s and D.mean_scalar(s) is s
    t = torch.zeros((4, 3), dtype=torch.float64)
    assert D.merged_adv_stats_epoch(t) is t
    assert D.init_from_env() is False or "RANK" in os.environ


def test_env_sharding_is_invariant_to_world_size():
    """Philox reset streams are keyed by GLOBAL env id (oracle = same integer arithmetic as the
    kernel): rank r of W owning envs [r*N/W, (r+1)*N/W) draws what a single process would."""
    from oracle import cref
    N = 64
    whole = cref.BatchEnv("cartpole", N, seed=9).reset()
    for W in (2, 4):
        parts = [cref.BatchEnv("cartpole", N // W, seed=9, env_base=r * (N // W)).reset() for r in range(W)]
        assert np.array_equal(np.concatenate(parts), whole)
