"""Launch the default GAE kernel a few times at one shape (for rocprofv3 --pmc passes).
Buffers rotate over 4 sets (805 MB at C2 > the 256 MiB Infinity Cache)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import ops
p = argparse.ArgumentParser(); p.add_argument("--T", type=int, default=128); p.add_argument("--N", type=int, default=65536)
p.add_argument("--iters", type=int, default=12); p.add_argument("--variant", type=int, default=None)
p.add_argument("--packed", action="store_true", help="the packed-flag form (ppo_gae_dense_packed): r, v, flag bits in, adv out")
p.add_argument("--done-rate", type=float, default=1 / 50); p.add_argument("--trunc-rate", type=float, default=1 / 500); a = p.parse_args()
T, N = a.T, a.N
g = torch.Generator(device="cuda").manual_seed(1234)
sets = []
for _ in range(4):
    r = torch.randn((T, N), device="cuda", generator=g); v = torch.randn((T, N), device="cuda", generator=g)
    ev = torch.randn((T, N), device="cuda", generator=g)
    d = (torch.rand((T, N), device="cuda", generator=g) < a.done_rate).int()
    tr = ((torch.rand((T, N), device="cuda", generator=g) < a.trunc_rate) & (d == 0)).int(); tr[T - 1] = 1 - d[T - 1]
    sets.append((r, v, ev, d, tr, torch.empty_like(r), ops.pack_flag_words(d, tr)))
torch.cuda.synchronize()
for i in range(a.iters):
    s = sets[i % 4]
    if a.packed:
        ops.gae_dense_packed(s[0], s[1], s[2], s[6], 0.99, 0.95, out=s[5], variant=a.variant)
    else:
        ops.gae_dense(*s[:5], 0.99, 0.95, out=s[5], variant=a.variant)
torch.cuda.synchronize()
print("n_trunc per launch:", int((sets[0][4] != 0).sum()))
