"""CPU suite: `--gpus N` is real.  bench.py started as ONE plain process with --gpus N > 1 becomes N ranks
(ppo_amd/launch.py) before any GPU call, the ranks rendezvous and a collective counts them; a failing rank
fails the job; RCCL runs refuse to pile ranks onto fewer GPUs; ranks that disagree about the next collective
raise instead of hanging (PPO_AMD_DIST_DEBUG); the early-stop decision is taken by all ranks together
(the reference decides on the max over ALL envs, src/ppo.py:758)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _clean_env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PPO_AMD_SELF_LAUNCHED"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_bench_gpus2_dry_goes_through_the_self_launcher():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=_clean_env(),
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["self_launched"] is True
    assert line["max_rank_seen_by_max_reduce"] == 1.0 and len(line["rank_devices"]) == 2


def test_bench_under_torchrun_style_env_checks_world_against_gpus():
    env = _clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_rccl_run_refuses_more_ranks_than_gpus():
    import torch
    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 7)], env=_clean_env(),
                       capture_output=True, text=True, timeout=240)
    assert p.returncode != 0
    assert f"need {have + 7} visible GPUs, found {have}" in p.stderr, p.stderr[-2000:]


def test_self_launch_returns_the_failing_ranks_status_and_ends_the_others():
    prog = textwrap.dedent("""
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)            # a rank that would wait for its dead peer for ever
    """)
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        from ppo_amd import launch
        t0 = time.time()
        rc = launch.self_launch(3, [sys.executable, "-c", {prog!r}])
        print("rc", rc, "fast", time.time() - t0 < 60)
    """)
    p = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=200)
    assert "rc 7 fast True" in p.stdout, (p.stdout, p.stderr[-2000:])


_DIST_PROG = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    from ppo_amd import dist as D
    assert D.init_from_env("gloo")
    r = D.rank()
    mode = sys.argv[1]
    if mode == "agree":
        # the training loop's early-stop decision: only rank 1's shard reached the reward
        stop = D.any_rank(r == 1, "early_stop_reward")
        mx = D.max_scalar(float("nan") if r == 0 else 3.5)
        nn = D.max_scalar(float("nan"))
        g = D.gather_rows(torch.tensor([float(r), 10.0 + r]))
        print("RESULT", r, stop, mx, nn != nn, g.tolist(), D.sequence_number(), flush=True)
    else:
        # a desynchronised job: rank 0 goes on to a gradient all-reduce, rank 1 left the loop and takes a barrier
        try:
            if r == 0:
                D.allreduce_flat(torch.zeros(4675))
            else:
                D.barrier("bench_exit")
            print("RESULT", r, "no error", flush=True)
        except RuntimeError as e:
            print("RESULT", r, "raised", str(e).replace("\\n", " "), flush=True)
    torch.distributed.destroy_process_group()
""") % ROOT


def _run_dist(mode, **env_extra):
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from ppo_amd import launch
        sys.exit(launch.self_launch(2, [sys.executable, "-c", {_DIST_PROG!r}, {mode!r}], timeout_s=150))
    """)
    return subprocess.run([sys.executable, "-c", code], env=_clean_env(**env_extra), capture_output=True, text=True,
                          timeout=200)


def test_decisions_that_end_the_loop_are_taken_by_all_ranks():
    p = _run_dist("agree")
    assert p.returncode == 0, p.stderr[-2000:]
    res = sorted(l for l in p.stdout.splitlines() if l.startswith("RESULT"))
    assert len(res) == 2
    for r, l in enumerate(res):
        assert l == f"RESULT {r} True 3.5 True [[0.0, 10.0], [1.0, 11.0]] 4", l


def test_mismatched_collectives_raise_instead_of_hanging():
    p = _run_dist("mismatch", PPO_AMD_DIST_DEBUG="1")
    res = sorted(l for l in p.stdout.splitlines() if l.startswith("RESULT"))
    assert len(res) == 2, (p.stdout, p.stderr[-2000:])
    for l in res:
        assert "raised" in l and "DIFFERENT collectives" in l and "grad_bucket" in l and "bench_exit" in l, l
