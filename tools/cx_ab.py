"""GPU: kao_solve on drifted single topics with and without KAO-CX (test tooling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(300, 6, 2000)]
for B, R, P in sizes:
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    for cx in (0, -1):
        for seed in (3, 4):
            r = kao.solve([t], seed=seed, stop_at_bound=1, time_limit_s=budget, use_cycles=cx)[0]
            tm = kao.last_solve_timing()
            print(f"{B}x{P} cx={'on' if cx == 0 else 'off'} seed {seed}: {r.status} {r.objective} / {r.upper_bound} gap {r.upper_bound - r.objective} "
                  f"t_best {tm['time_to_best']:.2f} launches {int(tm['launches'])} bound launches {int(tm.get('bound_launches', -1))}", flush=True)
