"""GPU (round 6): how early does the perturbed iterate round to the optimum?  kao_lp_round with the iteration count capped, drift seeds 1-3 of
the north-star size: fractional partitions, violations, objective against the certificate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for ds in (1, 2, 3):
    pt = sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], []), 0.2, ds)[0]
    eps = min(1e-4, 1.5 / (pt.n_partitions * pt.rf))
    cert = kao.lp_bound(pt)["bound"]
    for k in (40, 50, 60, 70, 80, 90, 100, 200):
        r = kao.lp_round(pt, pert=eps, tol=1e-10, max_iters=k)
        print(f"drift seed {ds} cap {k}: iterations {r['iterations']} status {r['status']} fractional {r['fractional']} violations {r['violations'][0]} objective {r['objective']} (certificate {cert}) rounding {r['ms_round']:.0f} ms", flush=True)
