"""GPU (round 5): kao_solve with KAO-LP on drifted single topics: status, incumbent, certificate, time to the proof."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
budget = float(os.environ.get("BUDGET", "4"))
seeds = [int(x) for x in os.environ.get("SEEDS", "3").split(",")]
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for B, R, P in shapes:
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, int(os.environ.get("DSEED", "1")))[0]
    for seed in seeds:
        t0 = time.perf_counter()
        extra = {"restarts": int(os.environ["RESTARTS"])} if os.environ.get("RESTARTS") else {}
        r = kao.solve([t], seed=seed, stop_at_bound=1, time_limit_s=budget, **extra)[0]
        dt = time.perf_counter() - t0
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        print(f"{B}x{P} seed {seed}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} t_best {tm['time_to_best']:.3f}s total {dt:.3f}s "
              f"launches {tm['launches']} bound launches {tm['bound_launches']} / {tm['bound_iters']} it, cx {tm['cx_calls']} ({tm['cx_gains']} gains), gens {tm['generations']}, "
              f"lp {tm['lp_solves']} solves / {tm['lp_iters']} it, rounded {lp['rounded']} adopted {lp['adopted']} fractional {lp['fractional_partitions']}", flush=True)
