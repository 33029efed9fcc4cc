import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
import benchmark as harness
harness.run(planner="device", attempts=1, obstacles=True, device=0, quiet=True)
pr = cProfile.Profile()
pr.enable()
res = harness.run(planner="device", attempts=10, obstacles=True, device=0, quiet=True)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
print(res["planning_times"])
