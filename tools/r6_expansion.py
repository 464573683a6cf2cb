"""GPU (round 6): the expansion scenario of tools/r6_scenarios.py (100 brokers added to 1000, 100,000 partitions, drifted) under longer limits;
everything kao_solve reports about its LP."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 5
P = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
add, nid = [], 1000
for r in range(20):
    for _ in range(per):
        add.append((nid, r)); nid += 1
t = sy.drift(sy.make_cluster(1000, 20, 1, P, 3, [], add), 0.2, 1)[0]
print("bounds", kao.derive_bounds(t))
b = kao.lp_bound(t)
print("lp_bound", {k: (v if not hasattr(v, "shape") else "...") for k, v in b.items()})
kao.solve([t], seed=1, max_launches=1)
for lim in (3.0, 10.0):
    t0 = time.perf_counter()
    r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=lim)[0]
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print(f"limit {lim}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} {time.perf_counter() - t0:.3f}s")
    print("  timing", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()}))
    print("  lp", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in lp.items()}), flush=True)
if os.environ.get("ROUND"):
    eps = min(1e-4, 1.5 / (t.n_partitions * t.rf))
    for salt in (0, 1, 2, 3):
        for pert in (eps, eps / 4, eps * 4):
            r = kao.lp_round(t, pert=pert, salt=salt, tol=1e-10, max_iters=250)
            print(f"lp_round salt {salt} pert {pert:.2e}: objective {r['objective']} violations {r['violations']} iterations {r['iterations']} status {r['status']} fractional {r['fractional']} over_inflow {r['over_inflow']}", flush=True)
