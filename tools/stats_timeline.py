"""Per-workgroup wall-clock marks of the one-launch episode statistics (k_sel_incr_ints) behind a C2 rollout: entry,
counts scanned, new slots applied to the LDS histogram, flushed, ticket taken; for the last workgroup: histogram read
back, statistics written.  Needs a library whose stats.hip was compiled with -DPPO_STATS_TIMELINE.
usage: stats_timeline.py variants/libstatstl.so   (development aid)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import torch
from ppo_amd import Agent, SyncVecEnv
from ppo_amd.envs import DeviceEnv
N, T = 65536, 128
probe = DeviceEnv("cartpole")
torch.manual_seed(0)
agent = Agent(probe.observation_space, probe.action_space, 0, 1, "cuda")
ve = SyncVecEnv(lambda: DeviceEnv("cartpole"), N, T, 0, 0.0, agent, seed=0)
for _ in range(40): ve.rollout()
ve._flush_stats()
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 8, np.uint64)
rc = raw.ppo_stats_timeline_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size)
assert rc == 0, rc
m = buf.reshape(256, 8).astype(np.float64) / 100.0
t0 = m[:, 0].min()
names = ["entry", "counts scanned", "new slots applied (LDS)", "flushed to the global histogram", "ticket taken"]
for k, nm in enumerate(names):
    c = m[:, k] - t0
    print(f"{nm:34s} first {c.min():6.2f}  median {np.median(c):6.2f}  last {c.max():6.2f} us")
last = int(np.argmax(m[:, 4]))
print(f"last workgroup ({last}): ticket {m[last, 4] - t0:.2f}, histogram read back {m[last, 5] - t0:.2f}, statistics written {m[last, 6] - t0:.2f} us")
d = np.diff(m[:, :5], axis=1)
print("per workgroup (medians): " + ", ".join(f"{names[k]} -> {names[k+1]}: {np.median(d[:, k]):.2f}" for k in range(4)))
