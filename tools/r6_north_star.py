"""GPU (round 6): kao_solve on the north-star topic and its relatives under the north-star's own budget and a generous one:
status, objective, certificate, wall time to results in host memory, what KAO-LP's rounding did (fractional partitions, KAO-CX
descents, further solves).  Usage: r6_north_star.py [quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
cases = [  # (brokers, racks, partitions, rf, drift, drift seed)
    (1000, 20, 100_000, 3, 0.2, 1), (1000, 20, 100_000, 3, 0.2, 2), (1000, 20, 100_000, 3, 0.2, 3)]
if not quick:
    cases += [(1000, 20, 100_000, 3, 0.4, 1), (1000, 20, 200_000, 3, 0.2, 1), (2000, 20, 100_000, 3, 0.2, 1), (1000, 20, 100_000, 4, 0.2, 1),
              (1000, 20, 30_000, 3, 0.2, 2), (500, 10, 10_000, 3, 0.3, 4)]
limits = [float(x) for x in os.environ.get("LIMITS", "1.0,3.0").split(",")]
warm = sy.north_star_topic("drift100k")
kao.solve([warm], seed=1, max_launches=1)
for (B, R, P, RF, dr, ds) in cases:
    t = sy.drift(sy.make_cluster(B, R, 1, P, RF, [], []), dr, ds)[0]
    kao.solve([t], seed=1, max_launches=1)
    for lim in limits:
        t0 = time.perf_counter()
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=lim)[0]
        dt = time.perf_counter() - t0
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        print(f"{B}x{P} rf {RF} drift {dr} seed {ds} limit {lim}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} "
              f"read back {tm['results_read_back']:.3f}s (call {dt:.3f}s) launches {tm['launches']} cx {tm['cx_calls']} lp solves {int(lp['solves'])} iterations {int(lp['iterations'])} "
              f"rounded {int(lp['rounded'])} adopted {int(lp['adopted'])} fractional {int(lp['fractional_partitions'])}", flush=True)
