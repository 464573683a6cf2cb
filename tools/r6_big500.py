import os, sys, time
sys.path.insert(0, "/root/repo")
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.drift(sy.make_cluster(1000, 20, 1, 500_000, 3, [], []), 0.2, 1)[0]
try:
    t0 = time.perf_counter(); b = kao.lp_bound(t); print("lp_bound", b["bound"], b["iterations"], b["status"], b["ms"], time.perf_counter() - t0)
except Exception as e:
    print("lp_bound EXCEPTION", repr(e))
os.environ["KAO_SOLVE_TRACE"] = "1"
r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=6.0)[0]
print(r.status, r.objective, r.upper_bound, kao.last_solve_lp())
