"""GPU: drifted single topics (tools/drift_scale.py sizes) with the search-price feedback and the elite launches switched
on and off -- what each buys inside a fixed budget (test tooling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
sizes = ((100, 5, 1000), (300, 6, 2000), (500, 10, 5000))
if len(sys.argv) > 2:
    sizes = sizes[:int(sys.argv[2])]
variants = {"base": dict(use_prices=-1, elite_period=-1), "prices": dict(elite_period=-1), "elite": dict(use_prices=-1), "both": dict()}
for B, R, P in sizes:
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    kao.solve([t], seed=1, max_launches=1)  # warm the arena cache
    for name, kw in variants.items():
        t0 = time.perf_counter()
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget, **kw)[0]
        dt = time.perf_counter() - t0
        tm = kao.last_solve_timing()
        print(f"B={B} P={P} {name:7s}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} "
              f"t_best {tm['time_to_best']:.2f}s launches {int(tm['launches'])} total {dt:.2f}s", flush=True)
