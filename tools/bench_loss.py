"""Achieved HBM GB/s of the fused loss kernels at the SURVEY 8d sizes (dispatch durations from
rocprofv3 --kernel-trace are the reference; this prints event-based times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import ops
for B in (8192, 1 << 20, 1 << 23):
    g = torch.Generator(device="cuda").manual_seed(1)
    old_lp = -torch.randn(B, device="cuda", generator=g).abs()
    new_lp = old_lp + 0.05 * torch.randn(B, device="cuda", generator=g)
    old_v = torch.randn(B, device="cuda", generator=g); new_v = old_v + 0.3 * torch.randn(B, device="cuda", generator=g)
    adv = torch.randn(B, device="cuda", generator=g)
    for _ in range(3): ops.loss_fwd_bwd(new_lp, old_lp, new_v, old_v, adv, 0.2, 1, 1, 0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.loss_fwd_bwd(new_lp, old_lp, new_v, old_v, adv, 0.2, 1, 1, 0); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ts)[10] * 1e3
    print(f"B={B}: stats+loss fwd+bwd {t:.1f} us (events, 4 launches) -> {32 * B / t / 1e3:.0f} GB/s of 32 B/sample")
