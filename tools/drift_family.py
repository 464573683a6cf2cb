"""GPU: a family of drifted single topics of >= 1000 partitions (different shapes and drift seeds): how many does one kao_solve
call prove optimal within the budget, and how far are the others from their certificate (test tooling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
shapes = [(50, 5, 1000), (100, 5, 1000), (120, 4, 1200), (150, 6, 1500), (90, 3, 1500), (200, 8, 1600), (200, 5, 2000), (250, 10, 2000),
          (300, 6, 2000), (160, 4, 2400), (400, 8, 3000), (250, 5, 4000)]
proven = 0
rows = []
for (B, R, P) in shapes:
    for dseed in (1, 2):
        t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, dseed)[0]
        t0 = time.perf_counter()
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]
        dt = time.perf_counter() - t0
        gap = r.upper_bound - r.objective
        proven += r.status == "OPTIMAL_PROVEN"
        rows.append(gap)
        print(f"B={B:4d} R={R:2d} P={P:5d} drift seed {dseed}: {r.status:15s} objective {r.objective} certificate {r.upper_bound} gap {gap} "
              f"({100.0 * gap / max(1, r.upper_bound):.3f} %) {dt:.2f} s", flush=True)
print(f"proven optimal {proven}/{len(rows)}; gaps of the others: {sorted(g for g in rows if g)}")
