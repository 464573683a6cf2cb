"""GPU (round 6): multi-topic solves around the LP's domain -- several large topics in one call, one huge topic among many small ones,
mixed replication factors, broker weights at scale.  Per call: statuses, the largest gap, seconds, the LP's counts."""
import os, sys, time
from collections import Counter
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)

def retag(ts, tag):
    for i, t in enumerate(ts): t.name = f"{tag}-{i:03d}"
    return ts

cases = []
cases.append(("20 topics x 5,000 partitions, 500 brokers, drifted", sy.drift(sy.make_cluster(500, 10, 20, 5000, 3, [], []), 0.2, 1), 10.0))
cases.append(("5 topics x 30,000 partitions, 1000 brokers, drifted", sy.drift(sy.make_cluster(1000, 20, 5, 30_000, 3, [], []), 0.2, 1), 10.0))
huge = sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], []), 0.2, 1)
small = retag(sy.drift(sy.make_cluster(1000, 20, 200, 100, 3, [], []), 0.2, 2), "small")
cases.append(("one 100,000-partition topic + 200 topics x 100 partitions, 1000 brokers, drifted", huge + small, 5.0))
mixed = retag(sy.drift(sy.make_cluster(600, 12, 4, 12_000, 2, [], []), 0.2, 1), "rf2") + retag(sy.drift(sy.make_cluster(600, 12, 4, 12_000, 3, [], []), 0.2, 2), "rf3") + retag(sy.drift(sy.make_cluster(600, 12, 2, 6_000, 5, [], []), 0.2, 3), "rf5")
cases.append(("10 topics of RF 2 / 3 / 5, 6,000-12,000 partitions, 600 brokers, drifted", mixed, 10.0))
rm = [7, 77, 177, 277, 377, 477]
cases.append(("8 topics x 8,000 partitions, 6 of 500 brokers decommissioned, drifted", sy.drift(sy.make_cluster(500, 10, 8, 8000, 3, rm, []), 0.15, 4), 10.0))
for name, ts, lim in cases:
    try:
        kao.solve(ts, seed=1, max_launches=1)
        t0 = time.perf_counter()
        rs = kao.solve(ts, seed=3, stop_at_bound=1, time_limit_s=lim, schedule=int(os.environ.get("SCHEDULE", "0")))
        dt = time.perf_counter() - t0
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        st = Counter(r.status for r in rs)
        gaps = [int(r.upper_bound - r.objective) for r in rs if r.status not in ("INFEASIBLE_PROVEN", "NO_FEASIBLE")]
        bad = 0
        for t, r in zip(ts, rs):
            if r.status in ("INFEASIBLE_PROVEN", "NO_FEASIBLE"): continue
            obj, viol = kao.evaluate_batch(t, np.asarray(r.assignment)[None])
            bad += int(np.asarray(viol)[0][0]) != 0 or int(obj[0]) != r.objective
        print(f"{name} (limit {lim:g} s): {dict(st)} largest gap {max(gaps) if gaps else None} sum of gaps {sum(gaps)} read back {tm['results_read_back']:.3f}s (call {dt:.3f}s) launches {tm['launches']} "
              f"lp solves {int(lp['solves'])} iterations {int(lp['iterations'])} adopted {int(lp['adopted'])} cx {tm['cx_calls']} | evaluator disagrees on {bad}", flush=True)
    except Exception as e:
        print(f"{name}: EXCEPTION {e!r}", flush=True)
