"""GPU (round 6): KAO_SOLVE_TRACE of the 20 x 5,000-partition call of tools/r6_scenarios2.py (where do the turns go?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ts = sy.drift(sy.make_cluster(500, 10, n, 5000, 3, [], []), 0.2, 1)
kao.solve(ts, seed=1, max_launches=1)
os.environ["KAO_SOLVE_TRACE"] = "1"
t0 = time.perf_counter()
rs = kao.solve(ts, seed=3, stop_at_bound=1, time_limit_s=float(sys.argv[2]) if len(sys.argv) > 2 else 10.0)
print("statuses", [r.status for r in rs], "seconds", time.perf_counter() - t0, kao.last_solve_lp(), kao.last_solve_timing())
