"""GPU: drifted topics of growing size -- incumbent, certificate, gap after a fixed budget (test tooling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
for B, R, P in ((100, 5, 1000), (300, 6, 2000), (500, 10, 5000), (500, 10, 10000), (1000, 20, 30000)):
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    t0 = time.perf_counter()
    r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]
    dt = time.perf_counter() - t0
    tm = kao.last_solve_timing()
    gap = r.upper_bound - r.objective
    print(f"B={B} R={R} P={P}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {gap} ({100.0 * gap / max(1, r.upper_bound):.3f} %) "
          f"closed-form {kao.upper_bound(t)} t_best {tm['time_to_best']:.2f}s launches {int(tm['launches'])} bound launches {tm['bound_launches']} iters {tm['bound_iters']} total {dt:.2f}s", flush=True)
