"""Data-parallel glue: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on
ROCm; "gloo" in CPU tests).

The rollout and GAE shard by env with no exchange (each rank owns N/world envs; Philox
streams are keyed by GLOBAL env id, so results do not depend on the sharding).  Only the
update exchanges data: one flattened-gradient all-reduce per optimiser step (the dense net
is ~5-9k parameters, i.e. a latency-bound 20-40 KB message -- bucketed into ONE collective
rather than one per parameter), three doubles of advantage statistics when normalising,
the KL scalar that drives the early-stop decision and the early_stop_reward decision (so that
every rank takes the same branch), and the episode-statistics histograms (job-wide percentiles).
The reference has no distributed code at all; nothing here mirrors it.

Every collective goes through `_collective`, which numbers it.  With PPO_AMD_DIST_DEBUG=1 each call
first exchanges (sequence number, tag, element count, dtype, op) and raises on every rank if the ranks
are not about to enter the SAME collective -- an ordering bug then asserts instead of hanging.
"""
from __future__ import annotations

import os
import zlib

import torch
import torch.distributed as td

_seq = 0                       # collectives issued by this process so far
_DEBUG = os.environ.get("PPO_AMD_DIST_DEBUG", "0") not in ("", "0")
_TAG_WORDS = 6                 # 48 bytes of tag text travel with the check
_DTYPES = {torch.float32: 1, torch.float64: 2, torch.int32: 3, torch.int64: 4, torch.uint8: 5, torch.bool: 6}
trace: list = []               # (seq, tag, numel) of every collective when debugging (tests read it)


def is_dist() -> bool:
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def world_size() -> int:
    return td.get_world_size() if is_dist() else 1


def rank() -> int:
    return td.get_rank() if is_dist() else 0


def sequence_number() -> int:
    return _seq


def init_from_env(backend: str | None = None) -> bool:
    """Initialise from torchrun's (or ppo_amd.launch.self_launch's) RANK / WORLD_SIZE / MASTER_* if present.
    Under RCCL every rank needs a GPU of its own: LOCAL_RANK must name a visible device (no modulo -- eight
    ranks on a one-GPU box are an error, not a silent pile-up); under gloo (rehearsals on one GPU) ranks share."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    if td.is_initialized():
        return True
    if backend is None:  # PPO_AMD_DIST_BACKEND=gloo: rehearse several ranks on ONE GPU (RCCL wants a GPU per rank)
        backend = os.environ.get("PPO_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        ndev = torch.cuda.device_count()
        if local_rank >= ndev:
            raise RuntimeError(f"rank {os.environ['RANK']}: LOCAL_RANK={local_rank} but only {ndev} GPU(s) are visible; "
                               "RCCL needs one GPU per rank (PPO_AMD_DIST_BACKEND=gloo rehearses several ranks on one GPU)")
        torch.cuda.set_device(local_rank)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    td.init_process_group(backend=backend)
    return True


def _gloo_on_device(t: torch.Tensor) -> bool:
    return t.is_cuda and td.get_backend() == "gloo"


def _host_under_rccl(t: torch.Tensor) -> bool:
    return (not t.is_cuda) and td.get_backend() == "nccl"


def _raw_all_reduce(t: torch.Tensor, op):
    """RCCL collectives are stream-ordered kernels.  gloo (only used to rehearse several ranks on ONE GPU)
    is given host tensors: its device-tensor path stages through the host on streams of its own, and with
    both ranks on one GPU the trainer hung now and then inside all_reduce (DESIGN.md 6 has what is known).
    PPO_AMD_GLOO_DEVICE_TENSORS=1 hands gloo the device tensor anyway (diagnosis only)."""
    if _gloo_on_device(t) and os.environ.get("PPO_AMD_GLOO_DEVICE_TENSORS", "0") != "1":
        h = t.cpu()                       # synchronous device -> host copy: the stream is drained
        td.all_reduce(h, op=op)
        t.copy_(h)
        return
    if _host_under_rccl(t):               # RCCL moves device memory only (host-stepped envs hand over numpy-backed tensors)
        d = t.cuda()
        td.all_reduce(d, op=op)
        t.copy_(d)
        return
    td.all_reduce(t, op=op)


def _check_same_collective(tag: str, t: torch.Tensor, op):
    """Debug: every rank publishes what it is about to do; all must agree."""
    world, me = td.get_world_size(), td.get_rank()
    text = tag.encode()[:8 * _TAG_WORDS].ljust(8 * _TAG_WORDS, b"\0")
    words = [int.from_bytes(text[8 * k:8 * k + 8], "little", signed=True) for k in range(_TAG_WORDS)]
    mine = [_seq, zlib.crc32(tag.encode()), t.numel(), _DTYPES.get(t.dtype, 0), int(op == td.ReduceOp.MAX)] + words
    rows = torch.zeros((world, len(mine)), dtype=torch.int64, device="cuda" if td.get_backend() == "nccl" else "cpu")
    rows[me] = torch.tensor(mine, dtype=torch.int64)
    td.all_reduce(rows, op=td.ReduceOp.SUM)
    rows = rows.cpu().tolist()
    if any(r[:5] != rows[0][:5] for r in rows):
        def show(r):
            name = b"".join(int(w).to_bytes(8, "little", signed=True) for w in r[5:]).rstrip(b"\0").decode(errors="replace")
            return f"seq {r[0]} '{name}' numel {r[2]} dtype {r[3]} max {r[4]}"
        raise RuntimeError("ranks are about to enter DIFFERENT collectives: "
                           + "; ".join(f"rank {k}: {show(r)}" for k, r in enumerate(rows)))


def _collective(t: torch.Tensor, op, tag: str, broadcast_src: int | None = None):
    """All-reduce `t` in place over all ranks (or broadcast it from `broadcast_src`); the one door every
    collective of this package goes through."""
    global _seq
    if _DEBUG:
        _check_same_collective(tag, t, op)
        trace.append((_seq, tag, t.numel()))
    _seq += 1
    if broadcast_src is None:
        _raw_all_reduce(t, op)
    elif _gloo_on_device(t) and os.environ.get("PPO_AMD_GLOO_DEVICE_TENSORS", "0") != "1":
        h = t.cpu()                       # see _raw_all_reduce
        td.broadcast(h, broadcast_src)
        t.copy_(h)
    elif _host_under_rccl(t):
        d = t.cuda()
        td.broadcast(d, broadcast_src)
        t.copy_(d)
    else:
        td.broadcast(t, broadcast_src)


def _all_reduce(t: torch.Tensor, tag: str = "sum"):
    """Sum over ranks, in place."""
    _collective(t, td.ReduceOp.SUM, tag)


def broadcast_params(module: torch.nn.Module, src: int = 0):
    """Rank `src`'s parameters on every rank, as ONE collective (flattened)."""
    if not is_dist():
        return
    flat = torch.cat([p.data.reshape(-1) for p in module.parameters()])
    _collective(flat, td.ReduceOp.SUM, "broadcast_params", broadcast_src=src)
    off = 0
    for p in module.parameters():
        n = p.numel()
        p.data.copy_(flat[off:off + n].view_as(p))
        off += n


def allreduce_grads(params):
    """Mean of gradients over ranks in ONE collective (flattened bucket)."""
    if not is_dist():
        return
    with_grad = [p for p in params if p.grad is not None]
    if not with_grad:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in with_grad])
    _all_reduce(flat, "grads")
    flat /= world_size()
    off = 0
    for p in with_grad:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def allreduce_flat(flat: torch.Tensor):
    """Mean over ranks of a gradient bucket that is ALREADY one flat tensor (the fused update's gradients are
    views of it): the collective alone, no concatenation and no copy back."""
    if not is_dist():
        return
    _all_reduce(flat, "grad_bucket")
    flat /= world_size()


def merged_adv_stats(stats3: torch.Tensor) -> torch.Tensor:
    """All-gather {count, mean, M2} and merge (Chan et al.) so that every rank normalises its
    minibatch with the statistics of the GLOBAL minibatch."""
    if not is_dist():
        return stats3
    # every rank's triple in its own row of a zero matrix, summed: an all-gather made of the one
    # collective that both RCCL and gloo run on device tensors
    rows = stats3.new_zeros((world_size(), 3))
    rows[rank()] = stats3
    _all_reduce(rows, "adv_stats")
    from .ops import merge_stats
    return merge_stats([(g[0], g[1], g[2]) for g in rows])


def merged_adv_stats_epoch(stats: torch.Tensor) -> torch.Tensor:
    """merged_adv_stats for the [num_mb, 3] table of ops.adv_stats_epoch: the statistics of ALL of an epoch's
    minibatches merged over ranks in ONE collective instead of one per minibatch (same merge, element by element)."""
    if not is_dist():
        return stats
    rows = stats.new_zeros((world_size(),) + tuple(stats.shape))
    rows[rank()] = stats
    _all_reduce(rows, "adv_stats_epoch")
    from .ops import merge_stats
    return merge_stats([(g[:, 0], g[:, 1], g[:, 2]) for g in rows]).t().contiguous()


def mean_scalar(x: torch.Tensor) -> torch.Tensor:
    if not is_dist():
        return x
    y = x.clone()
    _all_reduce(y, "mean_scalar")
    return y / world_size()


def sum_tensor(x: torch.Tensor, tag: str = "sum_tensor") -> torch.Tensor:
    """Sum of a tensor over ranks, in place (histograms, counters)."""
    if is_dist():
        _all_reduce(x, tag)
    return x


def gather_rows(x: torch.Tensor, tag: str = "gather_rows") -> torch.Tensor:
    """[world, *x.shape]: every rank's `x`, as one sum collective of a zero matrix with one's own row set."""
    if not is_dist():
        return x.unsqueeze(0)
    rows = x.new_zeros((world_size(),) + tuple(x.shape))
    rows[rank()] = x
    _all_reduce(rows, tag)
    return rows


def max_scalar(x: float, tag: str = "max_scalar") -> float:
    """Max of a host float over ranks (the whole job is as slow as its slowest rank).  NaN counts as -inf
    (a rank that has nothing to report yet), NaN on every rank stays NaN."""
    if not is_dist():
        return x
    dev = "cuda" if td.get_backend() == "nccl" else "cpu"
    v = float("-inf") if x != x else float(x)
    y = torch.tensor([v], dtype=torch.float64, device=dev)
    _collective(y, td.ReduceOp.MAX, tag)
    out = float(y.item())
    return float("nan") if out == float("-inf") and x != x else out


def barrier(tag: str = "barrier"):
    """Host-level barrier: a one-element all-reduce whose result is read back."""
    if not is_dist():
        return
    dev = "cuda" if td.get_backend() == "nccl" else "cpu"
    y = torch.zeros(1, dtype=torch.float32, device=dev)
    _all_reduce(y, tag)
    y.item()


def any_rank(flag: bool, tag: str = "any_rank") -> bool:
    """True on every rank if `flag` is true on at least one: a decision that changes which collectives come
    next (leaving the training loop) must be taken by all ranks or by none."""
    if not is_dist():
        return bool(flag)
    return max_scalar(1.0 if flag else 0.0, tag) > 0.0
