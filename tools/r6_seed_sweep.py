"""GPU (round 6): the north-star shape (1000 brokers x 100,000 partitions, RF 3, 20 % drift) over drift seeds 4 .. 15 and three other
rack / broker shapes, kao_solve with time_limit_s = 1.0: status, gap, seconds, LP iterations.  A distribution, not one instance."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
cases = [(1000, 20, 100_000, 3, 0.2, s) for s in range(4, 16)] + [(1000, 10, 100_000, 3, 0.2, 1), (1000, 40, 100_000, 3, 0.2, 1), (800, 16, 100_000, 3, 0.1, 1), (1000, 20, 100_000, 3, 0.05, 1)]
kao.solve([sy.north_star_topic("drift100k")], seed=1, max_launches=1)
secs, proven = [], 0
for (B, R, P, RF, dr, ds) in cases:
    t = sy.drift(sy.make_cluster(B, R, 1, P, RF, [], []), dr, ds)[0]
    kao.solve([t], seed=1, max_launches=1)
    t0 = time.perf_counter()
    r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=1.0)[0]
    dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    secs.append(tm["results_read_back"]); proven += r.status == "OPTIMAL_PROVEN"
    print(f"{B}x{P} racks {R} drift {dr} seed {ds}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} "
          f"read back {tm['results_read_back']:.3f}s (call {dt:.3f}s) lp iterations {int(lp['iterations'])} fractional {int(lp['fractional_partitions'])} cx {tm['cx_calls']}", flush=True)
secs.sort()
print(f"== {proven} of {len(cases)} OPTIMAL_PROVEN under time_limit_s = 1.0; seconds min {secs[0]:.3f} median {secs[len(secs) // 2]:.3f} max {secs[-1]:.3f} ==")
