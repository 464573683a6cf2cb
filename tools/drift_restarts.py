"""GPU experiment behind the per-size sawtooth period and the auto-restart cap: objective reached
on a large drifted topic within a fixed budget for different restarts / iterations per launch / lam_max / period_log2.

  python tools/drift_restarts.py [budget_s] [params]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy

kao.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
t = sy.drift(sy.make_cluster(300, 6, 1, 2000, 3, [], []), 0.2, 1)[0]
if len(sys.argv) > 2 and sys.argv[2] == "params":
    for lam_max in (3, 6, 12, 40):
        for period_log2 in (5, 8, 11, 14):
            r = kao.solve([t], seed=3, restarts=256, iters_per_launch=1024, stop_at_bound=1, time_limit_s=budget, lam_max=lam_max,
                          period_log2=period_log2, dual_iters=-1)[0]
            tm = kao.last_solve_timing()
            print(f"lam_max={lam_max} period_log2={period_log2}: {r.status} obj {r.objective} t_best {tm['time_to_best']:.3f} "
                  f"launches {int(tm['launches'])}", flush=True)
else:
    for restarts in (0, 4096, 1024, 256, 64):
        for ipl in (256, 1024):
            r = kao.solve([t], seed=3, restarts=restarts, iters_per_launch=ipl, stop_at_bound=1, time_limit_s=budget)[0]
            tm = kao.last_solve_timing()
            print(f"restarts={restarts} iters/launch={ipl}: {r.status} obj {r.objective} bound {r.upper_bound} "
                  f"t_best {tm['time_to_best']:.3f} launches {int(tm['launches'])}", flush=True)
