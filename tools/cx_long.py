"""GPU: one drifted topic with a longer budget, seeds 3..; prints status per seed (test tooling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P = (int(v) for v in sys.argv[1:4])
budget = float(sys.argv[4]); seeds = [int(v) for v in sys.argv[5:]] or [3]
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
for seed in seeds:
    for kw in (dict(), dict(use_prices=-1)):
        r = kao.solve([t], seed=seed, stop_at_bound=1, time_limit_s=budget, **kw)[0]
        tm = kao.last_solve_timing()
        print(f"{B}x{P} seed {seed} {kw}: {r.status} {r.objective} / {r.upper_bound} t_best {tm['time_to_best']:.2f} total {tm['returned']:.2f} launches {tm['launches']} bound launches {tm['bound_launches']}", flush=True)
