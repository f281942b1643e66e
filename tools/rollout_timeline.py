"""Where the whole-rollout kernel's fixed per-launch time goes: per-wave wall-clock marks (100 MHz counter) at kernel
entry, after the parameter load / LDS fill, before the loop, after the loop, after the last bootstrap pass, at the end.
Needs a library built with -DPPO_ROLLOUT_TIMELINE (tools/build_variant.sh timeline -DPPO_ROLLOUT_TIMELINE).
usage: rollout_timeline.py variants/libtimeline.so [kind] [joint]   (development aid; PPO_AMD_NET picks the network mode)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import torch
from ppo_amd import Agent, SyncVecEnv
from ppo_amd.envs import DeviceEnv
kind = sys.argv[2] if len(sys.argv) > 2 else "cartpole"
joint = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N, T = 65536, 128
probe = DeviceEnv(kind)
torch.manual_seed(0)
agent = Agent(probe.observation_space, probe.action_space, 0, joint, "cuda")
ve = SyncVecEnv(lambda: DeviceEnv(kind), N, T, 0, 0.0, agent, seed=0)
ve._device_episode_stats = lambda **kw: None
for _ in range(60): ve.rollout()
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(1024 * 8, np.uint64)
rc = raw.ppo_rollout_timeline_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size)
assert rc == 0, rc
m = buf.reshape(1024, 8)[:, :6].astype(np.float64) / 100.0            # us
t0 = m[:, 0].min()
names = ["entry", "params+LDS loaded", "state loaded, loop starts", "loop done", "last bootstrap pass done", "end"]
for k, nm in enumerate(names):
    c = m[:, k] - t0
    print(f"{nm:28s} first {c.min():8.2f}  median {np.median(c):8.2f}  last {c.max():8.2f} us after the first wave's entry")
d = np.diff(m, axis=1)
for k in range(5):
    print(f"  {names[k]} -> {names[k+1]}: median {np.median(d[:, k]):8.2f} us, max {d[:, k].max():8.2f}")
print(f"loop per step: median {np.median(d[:, 2]) / T:.3f} us; slowest wave's loop {d[:, 2].max():.1f} us, fastest {d[:, 2].min():.1f} us")
print(f"kernel span (first entry -> last end): {m[:, 5].max() - t0:.1f} us")
loop = d[:, 2]
blk = np.arange(1024) // 4
print("loop duration by workgroup index mod 8 (the XCD a workgroup lands on, round robin):",
      " ".join(f"{np.median(loop[blk % 8 == x]):.1f}" for x in range(8)))
print("loop duration by wave within the workgroup (its SIMD):", " ".join(f"{np.median(loop[np.arange(1024) % 4 == x]):.1f}" for x in range(4)))
print("spread inside one XCD group (mod 8 == 0): min %.1f median %.1f max %.1f" % (loop[blk % 8 == 0].min(), np.median(loop[blk % 8 == 0]), loop[blk % 8 == 0].max()))
order = np.argsort(loop)
print("slowest 8 waves:", [(int(w), round(float(loop[w]), 1)) for w in order[-8:]], " fastest 8:", [(int(w), round(float(loop[w]), 1)) for w in order[:8]])
