"""Child process of tests/test_gpu_dp_train.py: one rank of a data-parallel PPO.train run (or the
single-process run it must equal).  Started fresh by the test -- never imported by pytest -- with
RANK / WORLD_SIZE / MASTER_* in its environment; writes what it recorded to <out>.rank<r>."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ppo_amd import PPO  # noqa: E402
from ppo_amd import dist as D  # noqa: E402
from ppo_amd.envs import DeviceEnv  # noqa: E402


def main():
    out, num_envs, T, batch, updates = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    kind = sys.argv[6] if len(sys.argv) > 6 else "cartpole"
    joint = int(sys.argv[7]) if len(sys.argv) > 7 else 1
    early_stop_reward = float(sys.argv[8]) if len(sys.argv) > 8 and sys.argv[8] != "none" else None
    if os.environ.get("PPO_AMD_DP_WATCHDOG"):      # where is this rank when the job hangs?  (every thread's stack, then exit)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PPO_AMD_DP_WATCHDOG"]), exit=True)
    train_fn = lambda: DeviceEnv(kind)  # noqa: E731
    D.init_from_env()                       # backend from PPO_AMD_DIST_BACKEND (gloo: both ranks share the one GPU)
    rank, world = D.rank(), D.world_size()
    torch.manual_seed(100 + 7 * rank)       # different init per rank: broadcast_params must equalise
    ppo = PPO(0.99, 0.95, 1, 1, 0, joint, True, seed=2025)
    rec = {"world": world, "rank": rank, "steps": [], "buffers": None, "init": None, "metrics": [], "local_max": []}

    def hook(info):
        ve = ppo.vec_env
        if rec["buffers"] is None:          # first minibatch of update 0: the rollout buffers of update 0
            rec["buffers"] = {k: getattr(ve, k).detach().cpu().clone() for k in
                              ("states", "actions", "rewards", "done_flags", "trunc_flags", "values", "log_probs")}
            rec["buffers"]["ev_dense"] = torch.where(ve.trunc_flags > 0, ve._ev_dense, torch.zeros_like(ve._ev_dense)).cpu()
        flat = torch.cat([p.detach().reshape(-1) for p in ppo.agent.parameters()]).cpu()
        rec["steps"].append(dict(update=info["update"], epoch=info["epoch"], start=info["start"],
                                 idx=info["idx"].cpu() if len(rec["steps"]) == 0 else None,
                                 policy_loss=float(info["policy_loss"]), critic_loss=float(info["critic_loss"]),
                                 entropy=float(info["entropy"]), kl_div=float(info["kl_div"]), params=flat))

    ppo.minibatch_hook = hook
    orig = ppo.train_step

    def train_step(*a, **k):
        if rec["init"] is None:
            rec["init"] = {n: v.detach().cpu().clone() for n, v in ppo.agent.state_dict().items()}
            torch.manual_seed(555)          # same minibatch permutations in every run and on every rank
        r = orig(*a, **k)
        rec["metrics"].append({k_: v for k_, v in ppo.last_metrics.items() if not isinstance(v, str)})
        rec["local_max"].append(float(ppo.vec_env.local_max_ep_return))   # this rank's OWN shard (the logged one is job-wide)
        return r

    ppo.train_step = train_step
    ppo.train(train_fn, num_updates=updates, num_envs=num_envs, steps_per_env=T, num_epochs=2, batch_size=batch,
              critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5, learning_rate=3e-4,
              target_div=0.05, early_stop_reward=early_stop_reward)
    rec["updates_done"] = ppo.updates
    rec["collectives"] = D.sequence_number()
    torch.save(rec, f"{out}.rank{rank}")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
