"""GPU: one drifted topic under KAO_SOLVE_TRACE=1 (test tooling): one_trace.py B R P dseed [budget] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KAO_SOLVE_TRACE"] = "1"
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P, d = (int(v) for v in sys.argv[1:5])
budget = float(sys.argv[5]) if len(sys.argv) > 5 else 3.0
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 3
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, d)[0]
r = kao.solve([t], seed=seed, stop_at_bound=1, time_limit_s=budget)[0]
print(B, R, P, d, r.status, r.objective, r.upper_bound, kao.last_solve_timing())
