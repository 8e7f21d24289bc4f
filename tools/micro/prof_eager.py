import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--launch", "eager", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
