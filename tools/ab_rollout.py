"""A/B of two library builds on one box: the whole-rollout kernel at BASELINE configs[1] (or acrobot), settled clocks.
usage: ab_rollout.py <libA.so> <libB.so> [kind] [joint] [N]   (each measurement in a fresh process, A B A B)"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ppo_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    import torch
    from ppo_amd import Agent, SyncVecEnv
    from ppo_amd.envs import DeviceEnv
    kind, joint, N = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    torch.manual_seed(0)
    probe = DeviceEnv(kind)
    agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
    ve = SyncVecEnv(lambda: DeviceEnv(kind), N, 128, 0, 0.0, agent, seed=0)
    ve._device_episode_stats = lambda **kw: None
    for _ in range(250): ve.rollout()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ve.rollout()
    e1.record(); torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) * 10:.1f} us per rollout")
else:
    a, b = sys.argv[1], sys.argv[2]
    rest = (sys.argv[3:] + ["cartpole", "1", "65536"][len(sys.argv) - 3:])
    for rep in range(3):
        for lib in (a, b):
            out = subprocess.run([sys.executable, __file__, "--one", lib] + rest, capture_output=True, text=True, timeout=120)
            print(os.path.basename(lib), out.stdout.strip() or out.stderr.strip()[-300:], flush=True)
