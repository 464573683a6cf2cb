"""GPU (round 6): the centering exponent of KAO-LP (KAO_LP_SIGEXP) on the north star's relatives: kao_solve under a 3-s limit -- status, seconds,
LP solves (a second solve = the first rounded iterate was not at the certificate), iterations, fractional partitions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
exps = os.environ.get("EXPS", "3 6 8 10").split()
cases = [  # (name, brokers, racks, partitions, rf, drift, drift seed, bounds)
    ("cap+1", 1000, 20, 100_000, 3, 0.2, 1, {"rep_hi": 301}), ("drift 0.4", 1000, 20, 100_000, 3, 0.4, 1, None), ("200k", 1000, 20, 200_000, 3, 0.2, 1, None),
    ("2000 brokers", 2000, 20, 100_000, 3, 0.2, 1, None), ("rf 4", 1000, 20, 100_000, 4, 0.2, 1, None), ("30k seed 2", 1000, 20, 30_000, 3, 0.2, 2, None),
    ("500x10000", 500, 10, 10_000, 3, 0.3, 4, None), ("10 racks", 1000, 10, 100_000, 3, 0.2, 1, None), ("40 racks", 1000, 40, 100_000, 3, 0.2, 1, None),
    ("50 racks", 1000, 50, 100_000, 3, 0.2, 1, None), ("300k", 1000, 20, 300_000, 3, 0.2, 1, None), ("800 brokers", 800, 16, 100_000, 3, 0.1, 1, None), ("drift 0.05", 1000, 20, 100_000, 3, 0.05, 1, None), ("600x12000", 600, 12, 12_000, 3, 0.2, 1, None)]
kao.solve([sy.north_star_topic("drift100k")], seed=1, max_launches=1)
tot = {e: [0, 0.0, 0, 0] for e in exps}
only = os.environ.get('ONLY')
for (name, B, R, P, RF, dr, ds, bo) in cases:
    if only and name not in only.split(','): continue
    t = sy.drift(sy.make_cluster(B, R, 1, P, RF, [], [], **({"bounds_override": bo} if bo else {})), dr, ds)[0]
    kao.solve([t], seed=1, max_launches=1)
    for e in exps:
        if e == "default": os.environ.pop("KAO_LP_SIGEXP", None)      # the shipped rule (10; 24 on topics of more than 131,072 slots)
        else: os.environ["KAO_LP_SIGEXP"] = e
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=float(os.environ.get('LIMIT', '3.0')))[0]
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        tot[e][0] += r.status == "OPTIMAL_PROVEN"; tot[e][1] += tm["results_read_back"]; tot[e][2] += int(lp["solves"]); tot[e][3] += int(lp["iterations"])
        print(f"{name} exponent {e}: {r.status} gap {r.upper_bound - r.objective} read back {tm['results_read_back']:.3f}s lp solves {int(lp['solves'])} iterations {int(lp['iterations'])} "
              f"fractional {int(lp['fractional_partitions'])} cx {tm['cx_calls']} launches {tm['launches']}", flush=True)
for e in exps:
    print(f"== exponent {e}: {tot[e][0]} of {len(cases)} proven, {tot[e][1]:.2f} s in total, {tot[e][2]} LP solves, {tot[e][3]} iterations ==")
