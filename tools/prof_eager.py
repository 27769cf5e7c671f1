import cProfile, pstats, sys, runpy
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--eager"]
cProfile.run("runpy.run_path('bench.py', run_name='__main__')", "/tmp/prof.out")
