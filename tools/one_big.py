"""GPU: the drifted 1000 x 30000 topic with a longer budget (test tooling).  usage: one_big.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
t = sy.drift(sy.make_cluster(1000, 20, 1, 30000, 3, [], []), 0.2, 1)[0]
r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]
tm = kao.last_solve_timing()
print(f"1000 x 30000, {budget:.0f} s: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} "
      f"t_best {tm['time_to_best']:.2f}s K-bound launches {tm['bound_launches']} iterations {tm['bound_iters']}", flush=True)
