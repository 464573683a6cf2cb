"""GPU: option sweep of kao_solve on one drifted topic (test tooling)."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P = (int(v) for v in sys.argv[1:4])
budget = float(sys.argv[4])
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
grid = [dict(), dict(elite_period=8), dict(elite_period=128), dict(period_log2=11), dict(period_log2=15), dict(lam_max=20), dict(lam_max=80),
        dict(iters_per_launch=128), dict(iters_per_launch=1024)]
for kw in grid:
    out = []
    for seed in (3, 4, 5):
        r = kao.solve([t], seed=seed, stop_at_bound=1, time_limit_s=budget, **kw)[0]
        out.append(f"{r.objective}/{r.upper_bound}@{kao.last_solve_timing()['time_to_best']:.1f}")
    print(kw, out, flush=True)
