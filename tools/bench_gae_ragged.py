"""Dispatch durations of the ragged (reference-signature) GAE path at C2; run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import ops
T, N = 128, 65536
g = torch.Generator(device="cuda").manual_seed(1234)
r = torch.randn((T, N), device="cuda", generator=g); v = torch.randn((T, N), device="cuda", generator=g)
d = (torch.rand((T, N), device="cuda", generator=g) < 1 / 50).int()
tr = ((torch.rand((T, N), device="cuda", generator=g) < 1 / 500) & (d == 0)).int(); tr[T - 1] = 1 - d[T - 1]
E = int(((d + tr) > 0).sum())
ev = torch.randn(E, device="cuda", generator=g)
for _ in range(10):
    ops.gae_ragged(r, v, ev, d, tr, 0.99, 0.95)
torch.cuda.synchronize()
print("E =", E)
