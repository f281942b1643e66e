"""Data-parallel glue: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on
ROCm; "gloo" in CPU tests).

The rollout and GAE shard by env with no exchange (each rank owns N/world envs; Philox
streams are keyed by GLOBAL env id, so results do not depend on the sharding).  Only the
update exchanges data: one flattened-gradient all-reduce per optimiser step (the dense net
is ~5-9k parameters, i.e. a latency-bound 20-40 KB message -- bucketed into ONE collective
rather than one per parameter), three doubles of advantage statistics when normalising, and
the KL scalar that drives the early-stop decision (so that every rank takes the same branch).
The reference has no distributed code at all; nothing here mirrors it.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as td


def is_dist() -> bool:
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def world_size() -> int:
    return td.get_world_size() if is_dist() else 1


def rank() -> int:
    return td.get_rank() if is_dist() else 0


def init_from_env(backend: str | None = None) -> bool:
    """Initialise from torchrun's RANK / WORLD_SIZE / MASTER_* if present."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    if td.is_initialized():
        return True
    if backend is None:  # PPO_AMD_DIST_BACKEND=gloo: rehearse several ranks on ONE GPU (RCCL wants a GPU per rank)
        backend = os.environ.get("PPO_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    td.init_process_group(backend=backend)
    return True


def broadcast_params(module: torch.nn.Module, src: int = 0):
    if not is_dist():
        return
    flat = torch.cat([p.data.reshape(-1) for p in module.parameters()])
    if _gloo_on_device(flat):             # see _all_reduce
        h = flat.cpu()
        td.broadcast(h, src)
        flat.copy_(h)
    else:
        td.broadcast(flat, src)
    off = 0
    for p in module.parameters():
        n = p.numel()
        p.data.copy_(flat[off:off + n].view_as(p))
        off += n


def _gloo_on_device(t: torch.Tensor) -> bool:
    return t.is_cuda and td.get_backend() == "gloo"


def _all_reduce(t: torch.Tensor):
    """Sum over ranks, in place.  RCCL collectives are stream-ordered kernels.  gloo (only used to rehearse
    several ranks on ONE GPU) is given host tensors: its device-tensor path stages through the host on
    streams of its own, and the cross-stream waits that come with it deadlocked now and then when both
    ranks shared one GPU (both ranks inside all_reduce; tests/test_gpu_dp_train.py, tools/dp_two_ranks.sh)."""
    if _gloo_on_device(t):
        h = t.cpu()                       # synchronous device -> host copy: the stream is drained
        td.all_reduce(h, op=td.ReduceOp.SUM)
        t.copy_(h)
        return
    td.all_reduce(t, op=td.ReduceOp.SUM)


def allreduce_grads(params):
    """Mean of gradients over ranks in ONE collective (flattened bucket)."""
    if not is_dist():
        return
    with_grad = [p for p in params if p.grad is not None]
    if not with_grad:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in with_grad])
    _all_reduce(flat)
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
    _all_reduce(flat)
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
    _all_reduce(rows)
    from .ops import merge_stats
    return merge_stats([(g[0], g[1], g[2]) for g in rows])


def merged_adv_stats_epoch(stats: torch.Tensor) -> torch.Tensor:
    """merged_adv_stats for the [num_mb, 3] table of ops.adv_stats_epoch: the statistics of ALL of an epoch's
    minibatches merged over ranks in ONE collective instead of one per minibatch (same merge, element by element)."""
    if not is_dist():
        return stats
    rows = stats.new_zeros((world_size(),) + tuple(stats.shape))
    rows[rank()] = stats
    _all_reduce(rows)
    from .ops import merge_stats
    return merge_stats([(g[:, 0], g[:, 1], g[:, 2]) for g in rows]).t().contiguous()


def mean_scalar(x: torch.Tensor) -> torch.Tensor:
    if not is_dist():
        return x
    y = x.clone()
    _all_reduce(y)
    return y / world_size()


def max_scalar(x: float) -> float:
    """Max of a host float over ranks (the whole job is as slow as its slowest rank)."""
    if not is_dist():
        return x
    dev = "cuda" if td.get_backend() == "nccl" else "cpu"
    y = torch.tensor([x], dtype=torch.float64, device=dev)
    td.all_reduce(y, op=td.ReduceOp.MAX)
    return float(y.item())
