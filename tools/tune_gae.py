"""Time every (VEC, U) instantiation of the GAE kernel at a given shape (tuning aid).
Buffers are rotated over `--sets` independent copies so that repeated launches are not served
from the 256 MiB Infinity Cache (one C2 working set is 201 MB)."""
import argparse
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_amd import ops

p = argparse.ArgumentParser()
p.add_argument("--T", type=int, default=128)
p.add_argument("--N", type=int, default=65536)
p.add_argument("--sets", type=int, default=4)
p.add_argument("--iters", type=int, default=40)
p.add_argument("--variants", default="0,1,2,3,4,5")
p.add_argument("--stagger", type=int, default=-1, help="carve the 6 arrays of a set out of one buffer, each offset by this many extra bytes (-1: separate torch allocations)")
a = p.parse_args()
T, N = a.T, a.N
g = torch.Generator(device="cuda").manual_seed(1234)
sets = []
for _ in range(a.sets):
    r = torch.randn((T, N), device="cuda", generator=g)
    v = torch.randn((T, N), device="cuda", generator=g)
    ev = torch.randn((T, N), device="cuda", generator=g)
    d = (torch.rand((T, N), device="cuda", generator=g) < 1 / 50).int()
    tr = ((torch.rand((T, N), device="cuda", generator=g) < 1 / 500) & (d == 0)).int()
    tr[T - 1] = 1 - d[T - 1]
    out = torch.empty_like(r)
    if a.stagger >= 0:  # same data, placed at controlled relative offsets (HBM channel/bank interleave experiment)
        nbytes = T * N * 4
        pitch = nbytes + a.stagger
        big = torch.empty(6 * pitch + 256, dtype=torch.uint8, device="cuda")
        carved = []
        for i, src in enumerate((r, v, ev, d, tr, out)):
            view = big[i * pitch: i * pitch + nbytes].view(src.dtype).view(T, N)
            view.copy_(src)
            carved.append(view)
        keep = big
        r, v, ev, d, tr, out = carved
        sets.append((r, v, ev, d, tr, out, keep))
        continue
    sets.append((r, v, ev, d, tr, out))
bytes_ = 24 * T * N
res = {}
for variant in [int(x) for x in a.variants.split(",")]:
    vec = 2 if variant in (1, 4, 5, 6) else 1
    if N % vec:
        continue
    for i in range(3):
        s = sets[i % len(sets)]
        ops.gae_dense(*s[:5], 0.99, 0.95, out=s[5], variant=variant)
    torch.cuda.synchronize()
    times = []
    for i in range(a.iters):
        s = sets[i % len(sets)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gae_dense(*s[:5], 0.99, 0.95, out=s[5], variant=variant)
        e1.record()
        times.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) * 1e3 for x, y in times)
    med = ts[len(ts) // 2]
    res[variant] = {"median_us": med, "min_us": ts[0], "GBps_median": bytes_ / med / 1e3}
    print(f"variant {variant}: median {med:8.1f} us  min {ts[0]:8.1f} us  -> {bytes_ / med / 1e3:7.0f} GB/s "
          f"({100 * bytes_ / med / 1e3 / 8000:.1f}% of 8 TB/s)", flush=True)
# streaming ceiling with the same traffic mix (5 reads + 1 write of 4 B per element, 16 B/lane)
from ppo_amd import _lib
lib = _lib.load()
times = []
for i in range(a.iters + 3):
    s = sets[i % len(sets)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.ppo_stream_ceiling(*(x.data_ptr() for x in s[:6]), T * N, torch.cuda.current_stream().cuda_stream)
    e1.record()
    times.append((e0, e1))
torch.cuda.synchronize()
ts = sorted(x.elapsed_time(y) * 1e3 for x, y in times[3:])
res["stream_ceiling"] = {"median_us": ts[len(ts) // 2], "min_us": ts[0], "GBps_median": bytes_ / ts[len(ts) // 2] / 1e3}
print(f"streaming ceiling (same traffic mix, no scan): median {ts[len(ts)//2]:.1f} us -> {bytes_ / ts[len(ts)//2] / 1e3:.0f} GB/s")
print(json.dumps({"T": T, "N": N, "sets": a.sets, "results": res}))
