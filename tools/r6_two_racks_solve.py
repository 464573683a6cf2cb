"""GPU (round 6): kao_solve on two-rack clusters (the LP's late iterations lose primal feasibility there: tools/r6_two_racks_trace.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for (B, R, P, rf) in ((200, 2, 6000, 3), (600, 2, 10_000, 3), (600, 2, 50_000, 3), (1000, 2, 100_000, 3), (600, 2, 30_000, 2), (600, 2, 30_000, 4), (600, 1, 30_000, 3)):
    t = sy.drift(sy.make_cluster(B, R, 1, P, rf, [], []), 0.2, 1)[0]
    kao.solve([t], seed=1, max_launches=1)
    r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=3.0)[0]
    tm, lp = kao.last_solve_timing(), kao.last_solve_lp()
    print(f"{B} x {P}, {R} rack(s), RF {rf}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} in {tm['results_read_back']:.3f} s, launches {tm['launches']}, "
          f"lp solves {int(lp['solves'])} iterations {int(lp['iterations'])} adopted {int(lp['adopted'])} fractional {int(lp['fractional_partitions'])} cx {tm['cx_calls']}", flush=True)
