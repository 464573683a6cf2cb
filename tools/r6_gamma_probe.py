"""GPU (round 6): the perturbed solve's iteration count against the fraction of the way to the boundary a blocked step takes
(KAO_LP_GAMMA, read at lp_begin): kao_solve of the drifted 1000 x 100,000 topic (drift seeds 1..4) and of drift30k under time_limit_s = 1.0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
gammas = [g for g in os.environ.get("GAMMAS", "0.9995 0.95 0.9 0.8").split()]
topics = [("100k seed %d" % s, sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], []), 0.2, s)[0]) for s in [int(x) for x in os.environ.get("SEEDS", "1 2 3 4").split()]]
topics.append(("30k seed 1", sy.north_star_topic("drift30k")))
kao.solve([topics[0][1]], seed=1, max_launches=1)
for name, t in topics:
    kao.solve([t], seed=1, max_launches=1)
    for g in gammas:
        os.environ[os.environ.get("HOOK", "KAO_LP_GAMMA")] = g
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=1.0)[0]
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        print(f"{name} gamma {g}: {r.status} objective {r.objective} certificate {r.upper_bound} read back {tm['results_read_back']:.3f}s "
              f"lp iterations {int(lp['iterations'])} solves {int(lp['solves'])} fractional {int(lp['fractional_partitions'])} cx {tm['cx_calls']}", flush=True)
