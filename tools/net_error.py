"""Error of the whole-rollout kernel's values / log-probs against a float64 PyTorch forward of the
same parameters on the recorded states, for both network modes and for PyTorch's own f32 forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
from ppo_amd import Agent, SyncVecEnv, ops
from ppo_amd.envs import DeviceEnv
for kind in ("cartpole", "acrobot"):
    for scale in (1.0, 60.0):
        torch.manual_seed(0)
        probe = DeviceEnv(kind)
        agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
        with torch.no_grad():
            agent.policy.weight.mul_(scale)
            for p in agent.parameters():          # move away from the orthogonal init, like a trained net
                p.add_(0.05 * torch.randn_like(p))
        a64 = copy.deepcopy(agent).double()
        for mode, name in ((ops.NET_F32, "f32 mfma"), (ops.NET_F16_SPLIT, "f16 split")):
            ve = SyncVecEnv(lambda: DeviceEnv(kind, seed=1), 4096, 64, 0, 0.0, agent, seed=1)
            ve.net_mode = mode
            ve.rollout()
            st = ve.states.detach().flatten(0, 1)
            with torch.no_grad():
                lg64, v64 = a64.policy_and_values(st.double())
                lg32, v32 = agent.policy_and_values(st)
            lp64 = torch.log_softmax(lg64, 1).gather(1, ve.actions.flatten().long()[:, None])[:, 0]
            lp32 = torch.log_softmax(lg32, 1).gather(1, ve.actions.flatten().long()[:, None])[:, 0]
            ev, elp = (ve.values.flatten().double() - v64).abs(), (ve.log_probs.flatten().double() - lp64).abs()
            tv, tlp = (v32.double() - v64).abs(), (lp32.double() - lp64).abs()
            print(f"{kind:8s} head x{scale:4.0f} {name:9s}: value err max {ev.max():.2e} mean {ev.mean():.2e} "
                  f"(torch f32: {tv.max():.2e} / {tv.mean():.2e})  logp err max {elp.max():.2e} mean {elp.mean():.2e} "
                  f"(torch f32: {tlp.max():.2e} / {tlp.mean():.2e})  |v|max {v64.abs().max():.2f}")
