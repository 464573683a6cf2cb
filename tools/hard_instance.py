"""Time-to-bound on harder single-topic families (one MI355X): equality bands, massive rebalances, RF change."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy

kao.init(0)
fam = {
    "replace 3 of 1000 brokers, P=3000 (all bands equalities)": sy.make_cluster(1000, 20, 1, 3000, 3, [7, 77, 777], [(1000, 7), (1001, 17), (1002, 17)])[0],
    "double 100 -> 200 brokers, P=400": sy.make_cluster(100, 4, 1, 400, 3, [], [(100 + i, i % 4) for i in range(100)])[0],
    "grow 60 -> 90 brokers, P=300, 6 racks": sy.make_cluster(60, 6, 1, 300, 3, [], [(60 + i, i % 6) for i in range(30)])[0],
    "RF 2 -> 3, P=500, B=120": sy.make_cluster(120, 5, 1, 500, 2, [3, 4], [(120, 3), (121, 4)], new_rf=3)[0],
    "shrink 100 -> 70 brokers, P=350": sy.make_cluster(100, 5, 1, 350, 3, list(range(30)), [])[0],
}


def drifted(B, R, P, RF, frac, seed=1):
    """A balanced cluster whose replicas have drifted: `frac` of the slots moved to random brokers (the closed-form
    bound has a gap here; the certificate comes from K-bound)."""
    import numpy as np
    t = sy.make_cluster(B, R, 1, P, RF, [], [])[0]
    rng = sy.SplitMix64(0xD21F7 + seed)
    cur = np.array(t.current).copy()
    for _ in range(int(P * RF * frac)):
        p, k, nb = rng.below(P), rng.below(RF), rng.below(B)
        if nb not in cur[p]:
            cur[p, k] = nb
    t.current = cur
    return t


fam["drifted 20 %: B=60, P=200, 4 racks"] = drifted(60, 4, 200, 3, 0.2)
fam["drifted 20 %: B=120, P=400, 4 racks"] = drifted(120, 4, 400, 3, 0.2)
fam["drifted 20 %: B=300, P=2000, 6 racks"] = drifted(300, 6, 2000, 3, 0.2)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
for name, t in fam.items():
    kao.solve([t], seed=1, iters_per_launch=128, max_launches=1)  # warm the arena cache for this size
    t0 = time.perf_counter()
    r = kao.solve([t], seed=3, iters_per_launch=256, stop_at_bound=1, time_limit_s=budget)[0]
    dt = time.perf_counter() - t0
    tm = kao.last_solve_timing()
    print(f"{name}: {r.status} objective {r.objective} bound {r.upper_bound} (gap {r.upper_bound - r.objective}) "
          f"time_to_best {tm['time_to_best']:.4f}s launches {tm['launches']} total {dt:.3f}s")
