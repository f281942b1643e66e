"""Achieved HBM GB/s of the fused loss kernels at the SURVEY 8d sizes.  Run under
`rocprofv3 --kernel-trace` and read the dispatch durations (tools/prof_summary.py); this script only
drives the launches: `sets` rotating input sets so that at B = 2^23 the working set (235 MB per set)
does not sit in the 256 MiB Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import ops
sets = 4
for B in (8192, 1 << 20, 1 << 23):
    g = torch.Generator(device="cuda").manual_seed(1)
    data = []
    for _ in range(sets):
        old_lp = -torch.randn(B, device="cuda", generator=g).abs()
        new_lp = old_lp + 0.05 * torch.randn(B, device="cuda", generator=g)
        old_v = torch.randn(B, device="cuda", generator=g); new_v = old_v + 0.3 * torch.randn(B, device="cuda", generator=g)
        adv = torch.randn(B, device="cuda", generator=g)
        logits = torch.randn(B, 2, device="cuda", generator=g)
        act = torch.randint(0, 2, (B,), device="cuda", dtype=torch.int32, generator=g)
        data.append((new_lp, old_lp, new_v, old_v, adv, logits, act))
    for it in range(3 * sets):
        new_lp, old_lp, new_v, old_v, adv, logits, act = data[it % sets]
        ops.loss_fwd_bwd(new_lp, old_lp, new_v, old_v, adv, 0.2, 1, 1, 0)
        ops.loss_logits_fwd_bwd(logits, act, old_lp, new_v, old_v, adv, 0.2, 1, 1)
    torch.cuda.synchronize()
    print(f"B={B}: {3 * sets} launches each of ppo_adv_stats + ppo_loss_fwd_bwd (28 B/sample + 4 B stats pass) and "
          f"ppo_loss_logits_fwd_bwd A=2 (28 B read + 20 B written per sample)")
