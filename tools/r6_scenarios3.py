"""GPU (round 6): model variants at scale through kao_solve (3-s limit): few racks (1, 2, 3: the usual availability-zone set-ups),
another weight scheme, broker weights, a widened per-broker cap, RF 4 on three racks."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
cases = []
for R in (1, 2, 3):
    cases.append((f"{R} rack(s): 600 brokers x 50,000 partitions RF 3, drifted", sy.drift(sy.make_cluster(600, R, 1, 50_000, 3, [], []), 0.2, 1)[0]))
cases.append(("3 racks, RF 4: 600 brokers x 30,000, drifted", sy.drift(sy.make_cluster(600, 3, 1, 30_000, 4, [], []), 0.2, 1)[0]))
cases.append(("3 racks: 999 brokers x 100,000 RF 3, drifted", sy.drift(sy.make_cluster(999, 3, 1, 100_000, 3, [], []), 0.2, 1)[0]))
cases.append(("weights ((8, 1), (3, 2)): 1000 x 50,000, drifted", sy.drift(sy.make_cluster(1000, 20, 1, 50_000, 3, [], [], weights=((8, 1), (3, 2))), 0.2, 1)[0]))
t = sy.drift(sy.make_cluster(1000, 20, 1, 50_000, 3, [], []), 0.2, 1)[0]
rng = np.random.default_rng(5)
t.broker_w = rng.integers(0, 3, t.n_brokers).astype(np.int32); t.broker_wl = rng.integers(0, 2, t.n_brokers).astype(np.int32)
cases.append(("broker weights 0..2 / 0..1: 1000 x 50,000, drifted", t))
cases.append(("per-broker cap + 1 (config 5's rule): 1000 x 100,000, drifted", sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], [], bounds_override={"rep_hi": 301}), 0.2, 1)[0]))
kao.solve([sy.north_star_topic("drift100k")], seed=1, max_launches=1)
for name, t in cases:
    try:
        print(f"{name}: bounds {kao.derive_bounds(t)}", flush=True)
        kao.solve([t], seed=1, max_launches=1)
        t0 = time.perf_counter()
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=3.0)[0]
        dt = time.perf_counter() - t0
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        ok = "-"
        if r.status not in ("INFEASIBLE_PROVEN", "NO_FEASIBLE"):
            obj, viol = kao.evaluate_batch(t, np.asarray(r.assignment)[None]); ok = f"objective {int(obj[0])} violations {int(np.asarray(viol)[0][0])}"
        print(f"   {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} read back {tm['results_read_back']:.3f}s (call {dt:.3f}s) launches {tm['launches']} "
              f"lp solves {int(lp['solves'])} iterations {int(lp['iterations'])} adopted {int(lp['adopted'])} fractional {int(lp['fractional_partitions'])} cx {tm['cx_calls']} | evaluator: {ok}", flush=True)
    except Exception as e:
        print(f"   EXCEPTION {e!r}", flush=True)
