"""A/B/C... of several library builds on one box: the whole-rollout kernel at BASELINE configs[1] (or acrobot),
settled clocks, each measurement in a fresh process, round robin, three rounds.
usage: PPO_AMD_NET=<mode> ab_multi.py kind joint N lib1.so lib2.so ...   (development aid)"""
import os, subprocess, sys
if sys.argv[1] == "--one":
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
    for _ in range(200): ve.rollout()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ve.rollout()
    e1.record(); torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) * 10:.1f} us per rollout ({os.environ.get('PPO_AMD_NET', 'auto')})")
else:
    kind, joint, N = sys.argv[1:4]
    libs = sys.argv[4:]
    for rep in range(int(os.environ.get("AB_ROUNDS", "3"))):
        for lib in libs:
            out = subprocess.run([sys.executable, __file__, "--one", lib, kind, joint, N], capture_output=True, text=True, timeout=120)
            print(f"{os.path.basename(lib):28s}", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr.strip()[-300:], flush=True)
