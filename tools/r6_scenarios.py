"""GPU (round 6): operational scenarios at the north star's scale through kao_solve (time_limit_s = 3.0): decommission, expansion, a
replication-factor change, other replication factors, more brokers, more partitions.  Status, gap, seconds, what the LP did; every
result is checked for feasibility by the product's own evaluator (violations 0)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)

def rack_even(rm_per_rack, add_per_rack, B=1000, R=20):
    rng = sy.SplitMix64(99)
    rm, add, nid = [], [], B
    for r in range(R):
        rm += rng.sample([b for b in range(B) if b % R == r], rm_per_rack)
        for _ in range(add_per_rack):
            add.append((nid, r)); nid += 1
    return rm, add

cases = []
rm, _ = rack_even(2, 0)
cases.append(("decommission 40 of 1000 brokers (2 per rack), drifted", sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, rm, []), 0.2, 1)[0]))
_, add = rack_even(0, 5)
cases.append(("expansion by 100 brokers (5 per rack), drifted", sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], add), 0.2, 1)[0]))
rm, add = rack_even(2, 2)
cases.append(("replace 40 brokers, drifted", sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, rm, add), 0.2, 1)[0]))
cases.append(("RF 2 -> 3", sy.make_cluster(1000, 20, 1, 100_000, 2, [], [], new_rf=3)[0]))
cases.append(("RF 3 -> 2", sy.make_cluster(1000, 20, 1, 100_000, 3, [], [], new_rf=2)[0]))
cases.append(("RF 2, drifted", sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 2, [], []), 0.2, 1)[0]))
cases.append(("RF 5, drifted, 60,000 partitions", sy.drift(sy.make_cluster(1000, 20, 1, 60_000, 5, [], []), 0.2, 1)[0]))
cases.append(("3000 brokers x 100,000, drifted", sy.drift(sy.make_cluster(3000, 20, 1, 100_000, 3, [], []), 0.2, 1)[0]))
cases.append(("300 brokers x 300,000, drifted", sy.drift(sy.make_cluster(300, 10, 1, 300_000, 3, [], []), 0.2, 1)[0]))
kao.solve([sy.north_star_topic("drift100k")], seed=1, max_launches=1)
for name, t in cases:
    try:
        kao.solve([t], seed=1, max_launches=1)
        t0 = time.perf_counter()
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=3.0)[0]
        dt = time.perf_counter() - t0
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        obj, viol = kao.evaluate_batch(t, np.asarray(r.assignment)[None]) if r.status not in ("INFEASIBLE_PROVEN", "NO_FEASIBLE") else ([-1], [[-1]])
        print(f"{name}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} read back {tm['results_read_back']:.3f}s (call {dt:.3f}s) "
              f"launches {tm['launches']} lp solves {int(lp['solves'])} iterations {int(lp['iterations'])} fractional {int(lp['fractional_partitions'])} cx {tm['cx_calls']} | evaluator: objective {int(obj[0])} violations {int(np.asarray(viol)[0][0])}", flush=True)
    except Exception as e:
        print(f"{name}: EXCEPTION {e!r}", flush=True)
