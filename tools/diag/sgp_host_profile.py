"""Where the host time of the 1-scan scene-graph training step goes: cProfile over bench.py --workload sgp (the step is
host-bound: host_enqueue_ms_per_step == ms_per_step)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--workload", "sgp", "--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--no-kernel-timing"] + sys.argv[1:]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
