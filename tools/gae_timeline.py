"""Per-wave wall-clock marks of the packed-flag GAE kernel (k_gae_words) at [128, 65536]: entry, first block consumed,
loop done, stores landed.  Needs a library whose gae.hip was compiled with -DPPO_GAE_TIMELINE.
usage: gae_timeline.py variants/libgaetl.so [sets]   (development aid; sets = 4: HBM-cold inputs, 1: cache-resident)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import torch
from ppo_amd import ops
T, N = 128, 65536
sets = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = torch.Generator(device="cuda").manual_seed(1)
data = []
for _ in range(sets):
    r = torch.randn((T, N), device="cuda", generator=g); v = torch.randn((T, N), device="cuda", generator=g)
    ev = torch.randn((T, N), device="cuda", generator=g)
    d = (torch.rand((T, N), device="cuda", generator=g) < 1 / 22).int()
    tr = torch.zeros((T, N), dtype=torch.int32, device="cuda"); tr[T - 1] = 1 - d[T - 1]
    words = ops.pack_flag_words(d, tr) if hasattr(ops, "pack_flag_words") else None
    data.append((r, v, ev, d, tr, words))
if data[0][5] is None:
    raise SystemExit("ops.pack_flag_words missing: adapt this tool")
raw = ctypes.CDLL(_lib.LIB_PATH)
for rep in range(3 * sets):
    r, v, ev, d, tr, words = data[rep % sets]
    ops.gae_dense_packed(r, v, ev, words, 0.99, 0.95)
torch.cuda.synchronize()
buf = np.zeros(1024 * 4, np.uint64)
rc = raw.ppo_gae_timeline_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), buf.size)
assert rc == 0, rc
m = buf.reshape(1024, 4).astype(np.float64) / 100.0
t0 = m[:, 0].min()
for k, nm in enumerate(["entry", "first block consumed", "loop done (last store issued)", "stores landed"]):
    c = m[:, k] - t0
    print(f"{nm:32s} first {c.min():7.2f}  median {np.median(c):7.2f}  last {c.max():7.2f} us")
dd = np.diff(m, axis=1)
print("per wave: entry->first block %.2f, first block->loop done %.2f (7 more blocks: %.2f each), loop done->landed %.2f us (medians)" % (
    np.median(dd[:, 0]), np.median(dd[:, 1]), np.median(dd[:, 1]) / 7, np.median(dd[:, 2])))
print("span first entry -> last landed: %.2f us" % (m[:, 3].max() - t0))
blk = np.arange(1024) // 4
print("wave duration by workgroup index mod 8:", " ".join(f"{np.median((m[:, 3] - m[:, 0])[blk % 8 == x]):.2f}" for x in range(8)))
