"""bench.py -- env-steps/sec of the PPO rollout + GAE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]       (N > 1 without a launcher: starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch, the bracket the reference times as
`env_steps_per_sec` (reference src/ppo.py:544-560): a T=128-step rollout of the env batch -- ONE launch
of the whole-rollout kernel (ppo_rollout_fused: per time step the dense 64-64 tanh policy / value
network on the matrix cores, Philox action sampling, the fp64 env transition with auto-reset, and, for
truncated episodes, the bootstrap value of the terminal observation from a second network pass in the
same kernel) --, the episode-statistics kernel of that rollout (reference :404-417, inside the
reference's bracket too) and the GAE kernel (the trainer's form, ppo_gae_dense_packed, reading the two flag
bits per element the rollout kernel leaves).  Workload = BASELINE.json configs[1]: CartPole,
N = 65536 envs x T = 128 per GPU.  With --gpus N every rank runs that workload on its own envs
(weak scaling; the path has no data-path collective -- envs shard by batch, and the gradient
all-reduce belongs to the update, outside this bracket).

Network arithmetic: `value` is measured in the library's default mode (ops.NET_AUTO): f16-split MFMA
products (22-bit operands, f32 accumulate) when ppo_rollout_pack_params' split_diag puts the parameter
set inside the range where that mode measures as accurate as PyTorch's f32 forward
(tests/test_gpu_net_precision.py), three-piece bf16 splits (PPO_NET_BF16X3: operands exact, f32-grade at
every weight scale) otherwise; the bench's freshly initialised policy is inside it.  The same steps with
PPO_NET_BF16X3 forced are reported beside it (`value_f32_grade`) -- the figure to quote against the
reference's f32 arithmetic -- and with every product an exact f32 FMA (`value_with_exact_f32_products`).
`sustained`: >= --sustained-seconds of back-to-back steps in the default mode.

Clocks: after idling, this GPU needs ~40 ms of sustained work before its clocks settle, and the
VALU-bound rollout kernel follows them (tools/rollout_phase.py).  `value` is therefore
measured after --settle-steps (default 100) further untimed steps in front of the W warm-up steps --
the regime a trainer lives in; the same W + K steps started from an idle GPU are measured first and
reported as `value_from_idle_clocks`.

After the timed region the same workload runs through PPO.train_step itself (a few updates, joint and
split networks, 4 minibatches of T*N/4): `train_step` carries the reference's own three rates
(src/ppo.py:559-560, :637-639) as the trainer reports them, under --gpus N with the RCCL gradient
all-reduce in the loop.

Prints ONE JSON line (rank 0).  `roofline` is for the GAE kernel (the HBM-bound kernel the north
star sets a target for), timed with HIP events on the launch stream inside the timed region;
`rollout_kernel` describes the step's dominant kernel, `step_breakdown_us` every kernel of a step from
its dispatch timestamps.  `ranks_seen` / `rank_devices`: what the collectives saw.  `cpu_baseline` is the oracle's serial per-env
Python restatement of the reference's CPU path (oracle/pyref.py, kind "port"), timed on rank 0 at N=1
on a bounded sample, plus the restated torch-CPU GAE / loss loops at the full configs[1] shape.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # MI355X_MICROARCH.md: best measured float4 copy (SURVEY 8d asks for both fractions)
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense BF16/FP16 MFMA
GAE_BYTES_PER_ELEM = 12.25     # the trainer's GAE (ppo_gae_dense_packed): r, v and two flag BITS read, adv written; the
                               # end value is read only where trunc != 0 (4 B more for those cells).  The API form
                               # (ppo_gae_dense, two i32 flag arrays) moves 20; SURVEY.md 8d's dense count is 24.


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # Defaults: 100 warm-up rollouts, 50 timed.  After an idle period the GPU's clocks take ~40 ms of
    # sustained work to settle (tools/rollout_phase.py: the VALU-bound rollout kernel runs 508, 550, ...
    # 481 (30th), 447 (60th), 435 us (80th and on) -- the same curve after a 0.5 s pause with the same env
    # batch, so it is the clock / power ramp, not the episode phases); a training run lives in the
    # settled regime, and that is what the default invocation measures.
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--settle-steps", type=int, default=100,
                   help="untimed steps BEFORE the warm-up steps: ~50 ms of the same work, enough for the GPU's clocks to "
                        "settle after idling (DESIGN.md 7); 0 = none.  The line also carries the figure measured from an "
                        "idle GPU with the given --warmup only (value_from_idle_clocks)")
    p.add_argument("--env", default="cartpole", choices=["cartpole", "acrobot"])
    p.add_argument("--num-envs", type=int, default=65536, help="envs PER GPU")
    p.add_argument("--rollout-len", type=int, default=128)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-train-step", action="store_true")
    p.add_argument("--train-step-timeout", type=float, default=240.0)
    p.add_argument("--gae-event-every", type=int, default=8,
                   help="in every K-th step of the timed region all launches carry start/stop event pairs; the GAE "
                        "kernel's duration for `roofline` comes from those steps (0: none)")
    p.add_argument("--cpu-sample-envs", type=int, default=4096)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--sustained-seconds", type=float, default=2.0,
                   help="after the timed region: this many seconds of back-to-back steps, reported beside `value` "
                        "(clocks / thermals over seconds; something for a busy sampler to see); 0 = skip")
    p.add_argument("--dry", action="store_true",
                   help="plumbing only: launch / rendezvous / count the ranks, no GPU work (runs on a CPU box over gloo)")
    return p.parse_args()


def cpu_baseline(kind, n_envs, T, seed):
    """The reference's CPU path shape, restated (oracle/pyref.py): one Python env object per env,
    serial stepping, per-element tensor writes, T-iteration torch-CPU GAE; one core."""
    from oracle import pyref
    from ppo_amd import Agent
    from ppo_amd.envs import make_spaces, KINDS
    torch.manual_seed(seed)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        obs_space, act_space = make_spaces(KINDS[kind])
        agent = Agent(obs_space, act_space, conv_net=0, joint_net=1, device="cpu")
        rng = np.random.default_rng(seed)
        envs = [pyref.make_env(kind, np.random.default_rng(rng.integers(1 << 62))) for _ in range(n_envs)]
        t_states = torch.stack([torch.tensor(e.reset()[0]) for e in envs])

        def policy(s):
            a, lp, v, _ = agent.get_actions_and_values(s, None)
            return a, lp, v

        t0 = time.perf_counter()
        out = pyref.serial_rollout(envs, t_states, policy, T)
        with torch.no_grad():
            ev = agent.get_values(out["end_states"])
            pyref.gae_torch(out["rewards"], out["values"], ev, out["done"], out["trunc"], 0.99, 0.95)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthreads)
    return {"value": n_envs * T / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{kind} N={n_envs} envs x T={T} (one rollout + bootstrap forward + torch-CPU GAE), "
                      f"{dt:.1f} s on 1 core; per-env cost is constant, so N=65536 extrapolates linearly"}


def cpu_gae_loss_baseline(T, N, seed):
    """BASELINE.md's GAE / loss CPU microbench at the full configs[1] shape: the reference's torch-CPU
    loops restated (oracle/pyref.gae_torch = src/ppo.py:468-486; oracle/pyref.losses_torch =
    :498-525 + autograd backward), default torch threads, synthetic inputs of SURVEY.md 8d."""
    from oracle import pyref
    g = torch.Generator().manual_seed(1234)
    r, v = torch.randn(T, N, generator=g), torch.randn(T, N, generator=g)
    done = (torch.rand(T, N, generator=g) < 1 / 50).to(torch.int32)
    trunc = ((torch.rand(T, N, generator=g) < 1 / 500) & (done == 0)).to(torch.int32)
    trunc[-1] = 1 - done[-1]
    E = int((done + trunc - done * trunc).sum())
    ev = torch.randn(E, generator=g)
    pyref.gae_torch(r[:8], v[:8], ev, done[:8], trunc[:8] * 0, 0.99, 0.95)      # warm the allocator
    t0 = time.perf_counter()
    adv = pyref.gae_torch(r, v, ev, done, trunc, 0.99, 0.95)
    gae_ms = 1e3 * (time.perf_counter() - t0)
    B = T * N // 4                                                               # one minibatch of four
    old_lp = -torch.randn(B, generator=g).abs()
    new_lp = (old_lp + 0.05 * torch.randn(B, generator=g)).requires_grad_(True)
    old_v = torch.randn(B, generator=g)
    new_v = torch.randn(B, generator=g).requires_grad_(True)
    a = adv.flatten()[:B].contiguous()
    t0 = time.perf_counter()
    pl, vl, cf, kl = pyref.losses_torch(torch.exp(new_lp - old_lp), new_v, old_v, a, 0.2, norm_adv=1, clip_va_loss=1)
    (pl + 0.5 * vl).backward()
    loss_ms = 1e3 * (time.perf_counter() - t0)
    return {"gae_ms": gae_ms, "gae_shape": [T, N], "loss_fwd_bwd_ms": loss_ms, "loss_batch": B,
            "torch_threads": torch.get_num_threads()}


def train_step_rates(kind, N, T, world, seed, updates=12, skip=4):
    """The reference's own bracket: PPO.train_step at this workload (global num_envs = N * world,
    one epoch of four minibatches), joint and split networks; medians over the updates after the first `skip`
    (an update is ~3 ms: the first few sit on the clock ramp of DESIGN.md 7 like the hot loop's warm-up steps)."""
    from ppo_amd import PPO, ops
    from ppo_amd.envs import DeviceEnv
    out = {}
    for name, joint in (("joint", 1), ("split", 0)):
        torch.manual_seed(seed)
        ppo = PPO(0.99, 0.95, 1, 1, 0, joint, True, seed=seed)
        rows = []
        orig = ppo.train_step

        def step(*a, _orig=orig, _ppo=ppo, _rows=rows, **k):
            r = _orig(*a, **k)
            _rows.append(dict(_ppo.last_metrics))
            return r

        ppo.train_step = step
        ppo.train(lambda: DeviceEnv(kind), num_updates=updates, num_envs=N * world, steps_per_env=T, num_epochs=1,
                  batch_size=T * N * world // 4, critic_coef=0.5, entropy_coef=0.01, clip_ratio=0.2, max_grad_norm=0.5,
                  learning_rate=3e-4)
        med = lambda key: float(np.median([m[key] for m in rows[skip:]]))
        out[name] = {"train_step_env_steps_per_sec": med("utils/env_steps_per_sec"),
                     "grad_steps_per_sec": med("utils/grad_steps_per_sec"),
                     "global_steps_per_sec": med("utils/global_steps_per_sec"),
                     "net_mode_used": ops.NET_NAMES[ppo.vec_env.net_mode_used]}
        ppo.vec_env.close()
        del ppo
        torch.cuda.empty_cache()
    out["config"] = (f"{updates} updates, medians of the last {updates - skip}; num_epochs 1, batch_size T*N/4 (global), norm_adv 1, "
                     f"clip_va_loss 1; minibatch step = ppo_dense_update_grads (fused forward + loss + backward + weight gradients, net mode of the rollout); "
                     f"epoch shuffle = ppo_shuffle_indices, minibatch statistics = ppo_adv_stats_epoch (one pass per epoch), "
                     f"clip_grad_norm_ + Adam.step = ppo_clip_adam_step (one launch)")
    return out


def main():
    args = parse()
    if os.environ.get("PPO_AMD_BENCH_WATCHDOG"):   # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PPO_AMD_BENCH_WATCHDOG"]), exit=True)
    from ppo_amd import launch
    backend = os.environ.get("PPO_AMD_BENCH_BACKEND", "nccl")   # "gloo" only to rehearse N>1 on a 1-GPU box
    if launch.needs_self_launch(args.gpus):
        # `python bench.py --gpus N` as ONE plain process: become N ranks.  The parent starts them BEFORE any GPU
        # call of its own, relays their output (inherited stdout) and exits with the first non-zero child status.
        if not args.dry:
            launch.check_gpu_count(args.gpus, backend, f"bench.py --gpus {args.gpus}")
        sys.exit(launch.self_launch(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    from ppo_amd import dist as D
    if args.dry:
        return dry_run(args, D, world, rank)
    from ppo_amd import PPO, Agent, SyncVecEnv, ops, _lib
    lib = _lib.load()
    from ppo_amd.envs import DeviceEnv

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: ppo_amd has no CPU fallback")
    if world > 1:
        D.init_from_env(backend)            # RCCL: LOCAL_RANK must name a GPU of its own (raises otherwise)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    # proof that the collective backend saw every rank, and where each one sits
    ranks_seen = int(D.sum_tensor(torch.ones(1, dtype=torch.float32, device=dev if backend == "nccl" or world == 1 else "cpu"),
                                  "ranks_seen").item())
    devices = D.gather_rows(torch.tensor([float(dev.index)], device=dev if backend == "nccl" or world == 1 else "cpu"),
                            "rank_devices").reshape(-1).int().tolist()

    N, T, kind = args.num_envs, args.rollout_len, args.env
    torch.manual_seed(args.seed + rank)
    ppo = PPO(0.99, 0.95, 1, 1, 0, 1, True, seed=args.seed)
    probe = DeviceEnv(kind)
    ppo.agent = agent = Agent(probe.observation_space, probe.action_space, conv_net=0, joint_net=1, device="cuda")
    if backend == "nccl":
        D.broadcast_params(agent)   # same policy on every rank (RCCL); the rollout itself needs no collective
    ppo.vec_env = ve = SyncVecEnv(lambda: DeviceEnv(kind), N, T, 0, 0.0, agent, seed=args.seed, env_base=rank * N)

    gae_events = []
    step_no = [0]
    SAMPLE_SLOTS = 4                               # launches of a step are at most this many (three today)

    def hot_path_step(record):
        # In every K-th step of the timed region ALL launches of the step (statistics of the previous rollout, rollout,
        # GAE) carry a start/stop event pair, and the GAE launch's pair is what `roofline` reports.  A pair on the GAE
        # launch ALONE, right behind an un-timed rollout kernel, reads 1-2 us longer than the same dispatch under
        # rocprofv3 --kernel-trace (rounds 2-3 measured it that way: 20.9-22.6 us against the trace's 20.3); with its
        # neighbours timestamped too the two agree (20.3-20.7 us).  A pair costs ~5.7 us of the step it sits in, so
        # K = 8 (six sampled steps in 50, ~2 us per step on average) -- `value` includes that cost.
        step_no[0] += 1
        record = record and args.gae_event_every > 0 and step_no[0] % args.gae_event_every == 0
        record = record and SAMPLE_SLOTS * (len(gae_events) + 1) <= 128       # event slots 0..127 (128.. are the step breakdown's)
        if record:
            _lib.check(lib.ppo_prof_arm(SAMPLE_SLOTS * len(gae_events), SAMPLE_SLOTS), "ppo_prof_arm")
        ve.rollout()
        with torch.no_grad():
            if ve._ev_ready:                       # whole-rollout kernel wrote V(terminal obs) itself
                ev_dense = ve._ev_dense
            else:
                ev_dense = ppo._scatter_end_values(ve, agent.get_values(ve.end_states))
            if ve._ev_ready:                       # ... and the packed flag bits: the trainer's path (PPO._rollout_advantages)
                adv = ops.gae_dense_packed(ve.rewards, ve.values, ev_dense, ve._flag_words, 0.99, 0.95)
            else:
                adv = ops.gae_dense(ve.rewards, ve.values, ev_dense, ve.done_flags, ve.trunc_flags, 0.99, 0.95)
            if record:                             # GAE is the step's last launch: its events sit in the last slot taken
                gae_events.append(lib.ppo_prof_next() - 1)
                _lib.check(lib.ppo_prof_arm(0, 0), "ppo_prof_arm")
        return adv

    def barrier():
        torch.cuda.synchronize()
        D.barrier("bench_barrier")

    def timed_steps(record):
        for _ in range(args.warmup):
            hot_path_step(record)  # same code path as the timed steps, so one-time costs land here
        ve._flush_stats()
        barrier()
        gae_events.clear()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hot_path_step(record)
        ve._flush_stats()          # the last rollout's episode statistics belong to the bracket too (src/ppo.py:404-417)
        barrier()
        return time.perf_counter() - t0

    # From an idle GPU, with the given warm-up only: after idling this GPU needs ~40 ms of sustained work
    # before its clocks settle (tools/rollout_phase.py), so a short run sits on that ramp.  Reported beside
    # `value`, which is measured after --settle-steps more untimed steps: a trainer lives in that regime.
    hot_path_step(False)           # one-time costs (allocations, first launches) are not the clock ramp
    ve._flush_stats()
    barrier()
    time.sleep(0.5)
    elapsed_idle = timed_steps(False)
    for _ in range(args.settle_steps):
        hot_path_step(False)
    elapsed = timed_steps(True)
    mode_used = ops.NET_NAMES[ve.net_mode_used]

    def max_over_ranks(x):
        return D.max_scalar(x, "bench_time_max")

    elapsed = max_over_ranks(elapsed)
    elapsed_idle = max_over_ranks(elapsed_idle)

    def forced_mode_steps(mode):
        """The same K steps with the network arithmetic forced to `mode`."""
        ve.net_mode = mode
        for _ in range(2):
            hot_path_step(False)
        ve._flush_stats()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hot_path_step(False)
        ve._flush_stats()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        ve.net_mode = ops.NET_AUTO
        return dt

    # Beside `value` (the library's default: two-piece f16 splits where they measure as accurate as f32): the same
    # steps at the reference's arithmetic.  PPO_NET_BF16X3: operands represented exactly in three bf16 pieces, the
    # six largest cross terms of every product in one f32 accumulator -- as accurate as the exact-f32 kernels at
    # every weight scale (tests/test_gpu_net_precision.py), and what `auto` runs outside the f16 split's limits.
    # PPO_NET_F32: every product an exact f32 FMA on the f32 matrix-core path.
    elapsed_bf16x3 = forced_mode_steps(ops.NET_BF16X3)
    elapsed_f32 = forced_mode_steps(ops.NET_F32)

    # Sustained: >= --sustained-seconds of back-to-back steps in the default mode (clocks and thermals over seconds
    # instead of the 25 ms of the timed region; also something a once-per-second busy sampler can see)
    sustained = None
    if args.sustained_seconds > 0:
        n_sus = max(args.steps, int(np.ceil(args.sustained_seconds / (elapsed / args.steps))))
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            hot_path_step(False)
        ve._flush_stats()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        sustained = {"steps": n_sus, "seconds": dt, "value": n_sus * T * N * world / dt, "ms_per_step": 1e3 * dt / n_sus}

    # Where a step's time goes: 16 more back-to-back steps whose kernel dispatches all carry start/stop timestamps
    # (ppo_prof_arm: hipExtLaunchKernelGGL's events time the dispatch packet itself, like rocprofv3 --kernel-trace;
    # a torch.cuda.Event pair around a launch also times two barrier packets and reads 25-50 us high).  Outside the
    # timed region: an event pair costs ~5.7 us of the step it sits in.  A step launches, in this order: the
    # episode statistics of the PREVIOUS rollout (1 kernel: ppo_episode_stats_incremental_ints), the rollout kernel, GAE.
    us = C.c_float()

    def slot_us(slot):
        _lib.check(lib.ppo_prof_elapsed_us(slot, C.byref(us)), "ppo_prof_elapsed_us")
        return us.value

    for _ in range(4):
        hot_path_step(False)
    per_kernel = []
    bd_t0 = 0.0
    # launches per step, of which statistics kernels (the override is for A/B runs against an older library build)
    L, NS = (int(x) for x in os.environ.get("PPO_AMD_BENCH_STEP_LAUNCHES", "3,1").split(","))
    for k in range(16):
        if k == 4:                                 # (first four: settle)
            torch.cuda.synchronize()
            bd_t0 = time.perf_counter()
        _lib.check(lib.ppo_prof_arm(128 + L * k, L), "ppo_prof_arm")
        hot_path_step(False)
    _lib.check(lib.ppo_prof_arm(0, 0), "ppo_prof_arm")
    torch.cuda.synchronize()
    bd_wall_us = (time.perf_counter() - bd_t0) / 12 * 1e6     # wall time per step of the same twelve (instrumented) steps
    per_kernel = np.array([[slot_us(128 + L * k + j) for j in range(L)] for k in range(4, 16)])   # (first four: settle)
    stats_us, ro_us, gae_bd_us = per_kernel[:, :NS].sum(1).mean(), per_kernel[:, NS].mean(), per_kernel[:, NS + 1].mean()
    ro_ms = float(ro_us) * 1e-3
    waves = (N + 63) // 64 if N > 32768 else (N + 31) // 32          # 64- or 32-env waves (csrc/rollout.hip)
    mfma_per_wave_step = 60 if N > 32768 else 30                      # 32x32x16 f16 MFMAs, joint net
    ro_flops = mfma_per_wave_step * 2 * 32 * 32 * 16 * waves * T
    ro_bytes = (4 * probe.observation_space.shape[0] + 24.25) * N * T # states + action, log-prob, value, reward, 2 flags, flag bits

    gae_us = [slot_us(slot) for slot in gae_events]
    gae_ms = float(np.mean(gae_us)) * 1e-3 if gae_us else float("nan")
    n_trunc = float((ve.trunc_flags != 0).sum())   # truncated cells of the last rollout (4 B of end value each)
    gae_bytes = int(GAE_BYTES_PER_ELEM * T * N + 4 * n_trunc)
    achieved = gae_bytes / (gae_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r03_gae_packed_pmc_traffic.json")
    if os.path.exists(pmc):  # per-launch HBM bytes from rocprofv3 --pmc passes at this shape
        rec = json.load(open(pmc))
        if rec.get("T") == T and rec.get("N") == N:
            traffic = rec.get("hbm_bytes_per_launch")

    line = None
    if rank == 0:
        total_steps = args.steps * T * N * world
        line = {
            "metric": "env-steps/sec (rollout+GAE)", "value": total_steps / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64 env physics; f32 network, buffers, GAE (network products in the library's auto mode: two-piece "
                     "f16 splits of both operands, 22 bits each, f32 accumulate, chosen per parameter set only where that "
                     "measures as accurate as PyTorch's f32 forward -- tests/test_gpu_net_precision.py; three-piece bf16 "
                     "splits, f32-grade at every weight scale, otherwise: value_f32_grade is the whole run in that mode)",
            "data": "synthetic",
            "ranks_seen": ranks_seen, "rank_devices": devices, "backend": backend if world > 1 else None,
            "clock_settle_steps": args.settle_steps,
            "value_from_idle_clocks": total_steps / elapsed_idle,
            "ms_per_step_from_idle_clocks": 1e3 * elapsed_idle / args.steps,
            "value_f32_grade": total_steps / elapsed_bf16x3,
            "ms_per_step_f32_grade": 1e3 * elapsed_bf16x3 / args.steps,
            "value_with_exact_f32_products": total_steps / elapsed_f32,
            "ms_per_step_with_exact_f32_products": 1e3 * elapsed_f32 / args.steps,
            "sustained": sustained,
            "config": {"workload": f"{kind} batched stepping + GAE, N={N} envs/GPU x T={T}, dense 64-64 tanh policy "
                                   f"(joint net, f32) inside the whole-rollout kernel, BASELINE configs[1]",
                       "num_envs_per_gpu": N, "rollout_len": T, "net_mode": f"auto -> {mode_used}", "parallelism": f"env-sharded x{world}, no data-path collective"},
            "roofline": {"kernel": "k_gae_words (ppo_gae_dense_packed)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
                         "avg_launch_us": gae_ms * 1e3, "n_samples": len(gae_us), "algorithmic_bytes_per_launch": gae_bytes,
                         # the same kernel in the twelve instrumented steps BEHIND the timed region (step_breakdown_us);
                         # avg_launch_us above is from the sampled steps INSIDE it (hot_path_step), n_samples of them
                         "avg_launch_us_instrumented_steps": float(gae_bd_us),
                         "frac_instrumented_steps": gae_bytes / (float(gae_bd_us) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "achieved_at_api_20B_per_elem": 20 * T * N / (gae_ms * 1e-3) / 1e9,
                         "achieved_if_dense_24B_per_elem": 24 * T * N / (gae_ms * 1e-3) / 1e9,
                         "note": "algorithmic bytes = what this kernel has to move: 12.25 B/element (r, v, two flag bits "
                                 "in; adv out) + 4 B per truncated cell.  The same scan through the API's two i32 flag "
                                 "arrays (ppo_gae_dense) is 20 B/element and SURVEY.md 8d's dense convention 24: the two "
                                 "achieved_* fields price this launch's duration at those counts (work done per second "
                                 "in the reference's terms, not bytes moved).  Timed inside the pipeline: part of the "
                                 "input, written by the rollout kernel just before, is served by the 256 MiB Infinity "
                                 "Cache; with HBM-cold inputs (profiles/r03_gae_nt_store_ab.txt) the kernel takes "
                                 "21.2 us, and `traffic` is the L2-to-fabric byte count of such a cold launch"},
            "step_breakdown_us": {"episode_statistics_1_kernel": float(stats_us),
                                  "episode_statistics_each": [float(x) for x in per_kernel[:, :NS].mean(0)],
                                  "k_rollout": float(ro_us),
                                  "k_gae": float(gae_bd_us), "sum": float(stats_us + ro_us + gae_bd_us),
                                  "wall_per_step_of_these_steps": float(bd_wall_us),
                                  "note": "kernel durations from dispatch timestamps (ppo_prof_arm), means over 12 "
                                          "back-to-back steps after the timed region, with the wall time per step of those "
                                          "same twelve steps (every launch carrying an event pair, ~1 us each, so they run "
                                          "a little slower than the timed region's); wall minus sum is dispatch gaps "
                                          "between the three launches of a step"},
        }
        line["rollout_kernel"] = {
            "kernel": f"k_rollout (ppo_rollout_fused), {mode_used}", "bound": "valu-issue",
            "avg_launch_us": ro_ms * 1e3,
            "mfma_f16_tflops": ro_flops / (ro_ms * 1e-3) / 1e12, "mfma_f16_peak_tflops": MFMA_F16_PEAK_TFLOPS,
            "mfma_frac": ro_flops / (ro_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
            "hbm_write_gbs": ro_bytes / (ro_ms * 1e-3) / 1e9,
            # VALU issue: instructions per wave and step from the committed PMC summary (SQ_INSTS_VALU / waves / T,
            # CartPole joint, 64-env waves) against one wave64 instruction per SIMD every four cycles at 2.4 GHz
            "valu_insts_per_wave_step": (978 if (kind == "cartpole" and N > 32768) else None),
            "valu_issue_frac": ((978 * waves * T / (ro_ms * 1e-3)) / (1024 * 2.4e9 / 4)
                                if (kind == "cartpole" and N > 32768) else None),
            "note": "neither an HBM- nor an MFMA-bound kernel: by PMC (profiles/r03_rollout_pmc_f16_split.txt) "
                    "the VALU is active 73 % of the wave cycles (978 instructions per wave and step: tanh, operand "
                    "splits, heads, f64 physics, Philox), the 60 f16 MFMAs per wave and step are busy 25 %; duration "
                    "from the dispatch's own start/stop timestamps (step_breakdown_us)"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kind, args.cpu_sample_envs, T, args.seed)
            line["cpu_baseline"].update(cpu_gae_loss_baseline(T, N, args.seed))
    # the reference's own bracket, through the trainer (every rank takes part: gradient all-reduce)
    ve.close()
    del ve
    torch.cuda.empty_cache()
    status = 0
    if not args.no_train_step:
        # The headline line must not be lost to this extra section: an exception is reported inside the
        # line, and if the section does not finish in time (a rank stuck in a collective because another
        # one failed), a watchdog prints the line without it and ends the process -- with a NON-ZERO status on
        # every rank, so that the run is recorded as failed and a launcher ends the other ranks.
        import threading

        def give_up():
            msg = f"bench.py rank {rank}: train_step section did not finish within {args.train_step_timeout} s; giving up"
            if rank == 0:
                line["train_step"] = {"error": f"did not finish within {args.train_step_timeout} s"}
                print(json.dumps(line), flush=True)
            print(msg, file=sys.stderr, flush=True)
            os._exit(3)

        watchdog = threading.Timer(args.train_step_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            rates = train_step_rates(kind, N, T, world, args.seed)
        except Exception as e:
            rates = {"error": f"{type(e).__name__}: {e}"}
            print(f"bench.py rank {rank}: train_step section failed: {rates['error']}", file=sys.stderr, flush=True)
            status = 4
        D.barrier("bench_train_step_done")           # every rank is through before the watchdog is disarmed
        watchdog.cancel()
        if rank == 0:
            line["train_step"] = rates
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        D.barrier("bench_exit")
        torch.distributed.destroy_process_group()
    if status:
        sys.exit(status)


def dry_run(args, D, world, rank):
    """--dry: everything around the GPU work -- self-launch / torchrun environment, rendezvous, a collective that
    counts the ranks -- and nothing on the GPU.  Runs on a CPU box (gloo); tests/test_bench_launch.py uses it."""
    use_cuda = os.environ.get("PPO_AMD_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.is_available() \
        and torch.cuda.device_count() >= world
    backend = "nccl" if use_cuda else "gloo"
    if world > 1:
        D.init_from_env(backend)
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    ranks_seen = int(D.sum_tensor(torch.ones(1, device=dev), "ranks_seen").item())
    devices = D.gather_rows(torch.tensor([float(dev.index if use_cuda else -1)], device=dev), "rank_devices")
    slowest = D.max_scalar(float(rank), "bench_time_max")
    if rank == 0:
        print(json.dumps({"metric": "env-steps/sec (rollout+GAE)", "value": None, "unit": "env-steps/s", "n_gpus": world,
                          "ranks_seen": ranks_seen, "rank_devices": devices.reshape(-1).int().tolist(), "backend": backend,
                          "max_rank_seen_by_max_reduce": slowest, "dry": True, "steps": 0, "warmup": 0,
                          "self_launched": bool(os.environ.get("PPO_AMD_SELF_LAUNCHED"))}), flush=True)
    if world > 1:
        D.barrier("bench_exit")
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
