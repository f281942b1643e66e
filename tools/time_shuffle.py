"""ppo_shuffle_indices against torch.randperm at BASELINE configs[1]'s epoch size (128 x 65 536 rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 65536
out = torch.empty(n, dtype=torch.int64, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


k = [0]
def mine():
    k[0] += 1
    ops.shuffle_indices(n, 1, k[0], out=out)
print(f"n = {n}: ppo_shuffle_indices {timed(mine):.1f} us   torch.randperm {timed(lambda: torch.randperm(n, device='cuda')):.1f} us")
