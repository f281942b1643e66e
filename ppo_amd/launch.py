"""Self-launcher: `prog --gpus N` started as ONE plain process becomes N ranks, one per GPU.

The reference is a single process (src/ppo_cli.py:9-73; no launcher of any kind).  Data parallelism here is
one process per GPU over torch.distributed, normally started by torchrun, which sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*.  A program that is asked for N > 1 GPUs WITHOUT those variables calls `self_launch`
first thing: the parent starts N copies of itself with the variables set, waits, and exits with the first
non-zero child status (ending the others).  The parent never touches the GPU: it must not, because a process
that has initialised HIP cannot hand the device to children cleanly, and it has no work of its own.

Nothing here imports torch at module level on purpose; `torch.cuda.device_count()` is only used for the
early "not enough GPUs" check (it does not initialise the GPU on this stack).
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time

_RANK_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def under_launcher() -> bool:
    """True when this process is already one rank of a launched job (torchrun or self_launch)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def needs_self_launch(gpus: int) -> bool:
    return gpus > 1 and not under_launcher()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int:
    import torch
    return torch.cuda.device_count()       # counting devices does not initialise HIP on this image


def check_gpu_count(world: int, backend: str, what: str = "this run"):
    """RCCL wants one GPU per rank: fail before anything is launched instead of piling ranks onto one device."""
    if backend != "nccl":
        return
    have = visible_gpus()
    if have < world:
        raise SystemExit(f"{what}: {world} ranks over RCCL need {world} visible GPUs, found {have} "
                         f"(rehearse several ranks on one GPU with the gloo backend: PPO_AMD_DIST_BACKEND=gloo)")


def self_launch(world: int, argv: list[str] | None = None, env_extra: dict | None = None,
                poll_s: float = 0.05, timeout_s: float | None = None) -> int:
    """Run `argv` (default: this very command line) as `world` rank processes on this node and wait for them.
    Children inherit stdout / stderr, so rank 0's output IS this process's output.  Returns the exit status
    for the caller to exit with: 0 if every rank returned 0, else the first non-zero status seen (the
    remaining ranks are terminated by PID -- a rank stuck in a collective after its peer died would
    otherwise hang the job)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    if argv is None:
        argv = [sys.executable] + sys.argv
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PPO_AMD_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this stack (RCCL needs it)
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(argv, env=env))
    t0 = time.time()
    status = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc       # killed by signal n -> 128 + n
            if status != 0 or (timeout_s is not None and time.time() - t0 > timeout_s):
                if status == 0:
                    status = 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:                                       # exact PIDs we started, nothing by pattern
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        deadline = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return status
