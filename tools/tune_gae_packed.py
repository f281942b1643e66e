"""Time the packed-flag GAE kernels' instantiations (ppo_gae_dense_packed_variant: 0-2 column per lane, 3-5 time-split) next to the i32-flag default at one
shape, each launch timed by start/stop events on the dispatch itself (PPO_LAUNCH).  Buffers rotate over `--sets`
copies (4: every launch reads from HBM; 1: the 256 MiB Infinity Cache may serve part of it, as after the rollout
kernel in the trainer).  Checks every variant's output against the i32 form bit for bit."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ppo_amd import _lib, ops

p = argparse.ArgumentParser()
p.add_argument("--T", type=int, default=128)
p.add_argument("--N", type=int, default=65536)
p.add_argument("--sets", type=int, default=4)
p.add_argument("--iters", type=int, default=40)
p.add_argument("--variants", default="0,1,2,3,4,5")
p.add_argument("--trunc-rate", type=float, default=0.0, help="TimeLimit hits per element besides the forced last row")
a = p.parse_args()
T, N = a.T, a.N
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1234)
sets = []
for _ in range(a.sets):
    r = torch.randn((T, N), device="cuda", generator=g)
    v = torch.randn((T, N), device="cuda", generator=g)
    ev = torch.randn((T, N), device="cuda", generator=g)
    d = (torch.rand((T, N), device="cuda", generator=g) < 1 / 22).int()
    tr = ((torch.rand((T, N), device="cuda", generator=g) < a.trunc_rate) & (d == 0)).int()
    tr[T - 1] = 1 - d[T - 1]
    f8 = ops.pack_flag_words(d, tr)
    sets.append((r, v, ev, d, tr, f8, torch.empty_like(r)))


def elapsed(slot):
    us = ctypes.c_float()
    _lib.check(lib.ppo_prof_elapsed_us(slot, ctypes.byref(us)), "elapsed")
    return us.value


def time_it(fn):
    for i in range(3):
        fn(sets[i % len(sets)], None)
    torch.cuda.synchronize()
    ts = []
    for i in range(a.iters):
        fn(sets[i % len(sets)], i % 64)
        if i % 64 == 63 or i == a.iters - 1:
            ts += [elapsed(k) for k in range(i % 64 + 1)]
    ts.sort()
    return ts[len(ts) // 2], ts[0]


res = {}
want = [ops.gae_dense(*s[:5], 0.99, 0.95) for s in sets]
med, lo = time_it(lambda s, slot: ops.gae_dense(*s[:5], 0.99, 0.95, out=s[6], prof_slot=slot) if slot is not None
                  else ops.gae_dense(*s[:5], 0.99, 0.95, out=s[6]))
res["i32_default"] = {"median_us": med, "min_us": lo, "GBps_20B": 20 * T * N / med / 1e3}
print(f"i32 flags, default variant: median {med:6.1f} us  min {lo:6.1f}  -> {20 * T * N / med / 1e3:6.0f} GB/s at 20 B/element", flush=True)
for variant in [int(x) for x in a.variants.split(",")]:
    try:
        for s, w in zip(sets, want):
            got = ops.gae_dense_packed(s[0], s[1], s[2], s[5], 0.99, 0.95, variant=variant)
            assert torch.equal(got, w), f"variant {variant} differs"
    except _lib.PpoAmdError as e:
        print(f"packed variant {variant}: {e}")
        continue
    med, lo = time_it(lambda s, slot: ops.gae_dense_packed(s[0], s[1], s[2], s[5], 0.99, 0.95, out=s[6], variant=variant, prof_slot=slot))
    res[f"packed_{variant}"] = {"median_us": med, "min_us": lo, "GBps_12.25B": 12.25 * T * N / med / 1e3}
    print(f"packed variant {variant:2d}: median {med:6.1f} us  min {lo:6.1f}  -> {12.25 * T * N / med / 1e3:6.0f} GB/s at 12.25 B/element "
          f"({100 * 12.25 * T * N / med / 1e3 / 8000:.1f}% of 8 TB/s)", flush=True)
print(json.dumps({"T": T, "N": N, "sets": a.sets, "trunc_rate": a.trunc_rate, "results": res}))
