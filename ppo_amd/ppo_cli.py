"""CLI with the reference's positional arguments and options (src/ppo_cli.py:16-38).

    python -m ppo_amd.ppo_cli <experiment_name> <env_module> <num_updates> <num_envs> <steps_per_env>
        <num_epochs> <batch_size> <critic_coef> <entropy_coef> <clip_ratio> <max_grad_norm>
        <learning_rate> <discount_factor> <gae_factor> <norm_adv> <clip_va_loss> <conv_net>
        <joint_network> [--use_gpu 1] [--target_div F] [--render_every N] [--render_fps F]
        [--early_stop_reward F] [--checkpoint_path P --checkpoint_every N] [--resume_from P] [--gpus N]

The last four options are additions (the reference keeps no checkpoints and is a single process).  --gpus N > 1
started as one plain process becomes N data-parallel ranks, one per GPU (ppo_amd/launch.py; the same thing
`torchrun --nproc-per-node N -m ppo_amd.ppo_cli ...` does): num_envs and batch_size stay GLOBAL numbers.

`env_module` is looked up first as given (the reference's behaviour: any importable module
with a `train_fn`), then inside this package (env_cartpole, env_acrobot).  Under torchrun
the run is data-parallel over the launched ranks.
"""
import argparse
import importlib
import os
import sys

from . import dist as ppo_dist
from . import launch
from .ppo import PPO

_POSITIONAL = [
    ("experiment_name", str), ("env_module", str), ("num_updates", int), ("num_envs", int),
    ("steps_per_env", int), ("num_epochs", int), ("batch_size", int), ("critic_coef", float),
    ("entropy_coef", float), ("clip_ratio", float), ("max_grad_norm", float), ("learning_rate", float),
    ("discount_factor", float), ("gae_factor", float), ("norm_adv", int), ("clip_va_loss", int),
    ("conv_net", int), ("joint_network", int),
]


def build_parser():
    p = argparse.ArgumentParser(description="Train a PPO agent on the MI355X-native rollout/GAE/loss path.")
    for name, typ in _POSITIONAL:
        p.add_argument(name, type=typ)
    p.add_argument("--use_gpu", default=False, type=int)
    p.add_argument("--target_div", default=None, type=float)
    p.add_argument("--render_every", default=0, type=int)
    p.add_argument("--render_fps", default=0.0, type=float)
    p.add_argument("--early_stop_reward", default=None, type=float)
    p.add_argument("--checkpoint_path", default=None, type=str)
    p.add_argument("--checkpoint_every", default=0, type=int)
    p.add_argument("--resume_from", default=None, type=str)
    p.add_argument("--gpus", default=1, type=int)
    return p


def resolve_env_fn(name: str):
    try:
        mod = importlib.import_module(name)
    except ModuleNotFoundError:
        mod = importlib.import_module(f"{__package__}.{name}")
    return getattr(mod, "train_fn")


def main(argv=None):
    args = build_parser().parse_args(argv)
    if launch.needs_self_launch(args.gpus):            # before anything touches the GPU
        launch.check_gpu_count(args.gpus, os.environ.get("PPO_AMD_DIST_BACKEND") or "nccl", f"ppo_cli --gpus {args.gpus}")
        sys.exit(launch.self_launch(args.gpus, [sys.executable, "-m", "ppo_amd.ppo_cli"]
                                    + list(sys.argv[1:] if argv is None else argv)))
    if launch.under_launcher() and args.gpus > 1 and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"ppo_cli: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    print()
    ppo_dist.init_from_env()
    env_fn = resolve_env_fn(args.env_module)
    agent = PPO(args.discount_factor, args.gae_factor, args.norm_adv, args.clip_va_loss, args.conv_net,
                args.joint_network, args.use_gpu, project_name=args.experiment_name)
    agent.train(env_fn, args.num_updates, args.num_envs, args.steps_per_env, args.num_epochs, args.batch_size,
                args.critic_coef, args.entropy_coef, args.clip_ratio, args.max_grad_norm, args.learning_rate,
                args.target_div, args.render_every, args.render_fps, args.early_stop_reward,
                checkpoint_path=args.checkpoint_path, checkpoint_every=args.checkpoint_every,
                resume_from=args.resume_from)
    return agent


if __name__ == "__main__":
    main()
