"""GAE dispatch durations at the small BASELINE shapes (C4: T=128, N=64; C5: T=256, N=4096) and C2;
run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import ops
g = torch.Generator(device="cuda").manual_seed(1234)
for T, N in ((128, 64), (256, 4096), (128, 65536)):
    r = torch.randn((T, N), device="cuda", generator=g); v = torch.randn((T, N), device="cuda", generator=g)
    ev = torch.randn((T, N), device="cuda", generator=g)
    d = (torch.rand((T, N), device="cuda", generator=g) < 1 / 50).int()
    tr = ((torch.rand((T, N), device="cuda", generator=g) < 1 / 500) & (d == 0)).int(); tr[T - 1] = 1 - d[T - 1]
    for _ in range(10):
        ops.gae_dense(r, v, ev, d, tr, 0.99, 0.95)
    torch.cuda.synchronize()
