"""A/B of several library builds through bench.py itself (whole step: statistics + rollout + GAE), each in a fresh
process, round robin.  usage: ab_bench.py lib1.so lib2.so ... [-- extra bench args]   (development aid)"""
import json, os, subprocess, sys
if sys.argv[1] == "--one":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from ppo_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    sys.argv = [os.path.join(root, "bench.py")] + sys.argv[3:]
    import runpy
    runpy.run_path(sys.argv[0], run_name="__main__")
else:
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        extra = args[args.index("--") + 1:]
        args = args[:args.index("--")]
    for rep in range(int(os.environ.get("AB_ROUNDS", "3"))):
        for lib in args:
            out = subprocess.run([sys.executable, __file__, "--one", lib, "--no-cpu-baseline", "--no-train-step",
                                  "--sustained-seconds", "0"] + extra, capture_output=True, text=True, timeout=300)
            line = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
            if not line:
                print(os.path.basename(lib), "FAILED", out.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            b = d["step_breakdown_us"]
            print(f"{os.path.basename(lib):24s} {d['ms_per_step'] * 1e3:7.1f} us/step  f32-grade {d['ms_per_step_f32_grade'] * 1e3:7.1f}  "
                  f"gae(timed region) {d['roofline']['avg_launch_us']:5.2f}  breakdown: stats {next(v for k, v in b.items() if k.startswith('episode_statistics_') and k != 'episode_statistics_each'):5.1f} "
                  f"rollout {b['k_rollout']:6.1f} gae {b['k_gae']:5.2f}", flush=True)
