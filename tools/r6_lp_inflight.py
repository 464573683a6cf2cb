"""GPU (round 6): LP solves in flight (KAO_LP_MAX_RUNNING) on twenty drifted topics of 5,000 partitions and ten of 12,000 in one call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
sets = [("20 x 5,000 on 500 brokers", sy.drift(sy.make_cluster(500, 10, 20, 5000, 3, [], []), 0.2, 1)), ("10 x 12,000 on 600 brokers", sy.drift(sy.make_cluster(600, 12, 10, 12_000, 3, [], []), 0.2, 1))]
for name, ts in sets:
    kao.solve(ts, seed=1, max_launches=1)
    for k in (2, 4, 8, 16):
        os.environ["KAO_LP_MAX_RUNNING"] = str(k)
        rs = kao.solve(ts, seed=3, stop_at_bound=1, time_limit_s=20.0)
        tm, lp = kao.last_solve_timing(), kao.last_solve_lp()
        print(f"{name}, {k:2d} in flight: proven {sum(r.status == 'OPTIMAL_PROVEN' for r in rs)} of {len(rs)} in {tm['results_read_back']:.3f} s, {tm['launches']} launches, {int(lp['solves'])} LP solves, {tm['cx_calls']} KAO-CX calls", flush=True)
