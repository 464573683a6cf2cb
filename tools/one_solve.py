"""GPU: drifted topics, several solver seeds, one line each (test tooling): one_solve.py B R P dseed seeds(csv) [budget]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P, d = (int(v) for v in sys.argv[1:5])
seeds = [int(v) for v in sys.argv[5].split(",")]
budget = float(sys.argv[6]) if len(sys.argv) > 6 else 3.0
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, d)[0]
for sd in seeds:
    r = kao.solve([t], seed=sd, stop_at_bound=1, time_limit_s=budget)[0]
    tm = kao.last_solve_timing()
    print(f"B={B} P={P} d{d} seed {sd}: {r.status} obj {r.objective} cert {r.upper_bound} gap {r.upper_bound - r.objective} t_best {tm['time_to_best']:.2f} "
          f"launches {tm['launches']} cx {tm['cx_calls']}/{tm['cx_gains']} (+{tm['cx_further_starts']}) gens {tm['generations']}", flush=True)
