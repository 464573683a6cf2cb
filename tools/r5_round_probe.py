"""GPU (round 5): KAO-LP's primal side -- the perturbed LP's iterate rounded into an assignment (kao_lp_round) against the certificate
(kao_lp_bound).  Usage: r5_round_probe.py BxRxP[:seed] ...   (env SALTS=0,1  PERT=<eps>)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
salts = [int(x) for x in os.environ.get("SALTS", "0").split(",")]
pert = float(os.environ.get("PERT", "0"))
for spec in sys.argv[1:]:
    dims, _, seed = spec.partition(":")
    B, R, P = [int(x) for x in dims.split("x")]
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, int(seed or 1))[0]
    lb = kao.lp_bound(t)
    for salt in salts:
        t0 = time.perf_counter(); r = kao.lp_round(t, pert=pert, salt=salt); wall = time.perf_counter() - t0
        print(f"{B}x{P} d{seed or 1} salt {salt} pert {r['pert']:.2e}: objective {r['objective']} violations {r['violations'][0]} certificate {lb['bound']} "
              f"(LP {lb['iterations']} it {lb['ms']:.0f} ms) | perturbed LP {r['iterations']} it status {r['status']} {r['ms_lp']:.0f} ms, rounding {r['ms_round']:.1f} ms, "
              f"fractional {r['fractional']}, over inflow {r['over_inflow']}, wall {wall:.2f} s", flush=True)
