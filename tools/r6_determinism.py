"""GPU (round 6): same input, same seed, same answer -- for the paths this round added (K-init, several LPs in flight, the rack repair):
every call twice, assignments / objectives / certificates / counts compared."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
rng = sy.SplitMix64(sy.CONFIG_SEED + 5)
rm = rng.sample(list(range(1000)), 50)
add5 = [(1000 + i, b % 20) for i, b in enumerate(rm)]
addx = [(1000 + 5 * r + i, r) for r in range(20) for i in range(5)]
cases = [("20 x 5,000 (four LPs in flight)", sy.drift(sy.make_cluster(500, 10, 20, 5000, 3, [], []), 0.2, 1), dict(time_limit_s=20.0)),
         ("config 5 as one topic (K-init)", sy.make_cluster(1000, 20, 1, 100_000, 3, rm, add5, bounds_override={"rep_hi": 301}), dict(time_limit_s=5.0, restarts=64, iters_per_launch=128)),
         ("expansion by 100 brokers (rack repair)", sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], addx), 0.2, 1), dict(time_limit_s=5.0)),
         ("RF 2 -> 3 at 100,000 (100,000 holes)", sy.make_cluster(1000, 20, 1, 100_000, 2, [], [], new_rf=3), dict(time_limit_s=5.0))]
for name, ts, kw in cases:
    runs = []
    for _ in range(2):
        rs = kao.solve(ts, seed=3, stop_at_bound=1, **kw)
        tm, lp = kao.last_solve_timing(), kao.last_solve_lp()
        runs.append(([np.asarray(r.assignment).tobytes() for r in rs], [(r.status, int(r.objective), int(r.upper_bound)) for r in rs],
                     (tm["launches"], tm["search_iters"], tm["cx_calls"], tm["bound_launches"], int(lp["solves"]), int(lp["iterations"]), int(lp["adopted"]))))
    same = runs[0] == runs[1]
    print(f"{name}: identical {same}; statuses {sorted(set(s for s, _, _ in runs[0][1]))}; counts {runs[0][2]} / {runs[1][2]}", flush=True)
