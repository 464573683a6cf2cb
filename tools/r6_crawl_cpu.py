"""CPU look at the perturbed solve's slow stretch (oracle/kao_lp_port.c): trace of a drifted topic, perturbation as in kao_solve."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import numpy as np
import kao_oracle as ko, kao_lp as kl
from kafka_assignment_optimizer_amd import synthetic as sy

B, R, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pt = sy.drift(sy.make_cluster(B, R, 1, P, 3, list(range(0, B, 20)), [(B + i, (i * 20) % R) for i in range(len(range(0, B, 20)))]), 0.2, seed)[0]
ot = ko.Topic(name=pt.name, broker_ids=pt.broker_ids, rack_of=pt.rack_of, n_racks=pt.n_racks, n_partitions=pt.n_partitions,
              rf=pt.rf, current=pt.current, weights=pt.weights, bounds_override=dict(pt.bounds_override))
pert = min(1e-4, 1.5 / (P * 3))
for pr, tol in (((pert, 1e-10),) if os.environ.get('ONLY_PERT') else ((0.0, 1e-7), (pert, 1e-10))):
    t0 = time.time()
    r = kl.port_solve(ot, tol=tol, maxit=200, pert=pr, salt=0)
    print(f"pert {pr:g}: status {r['status']} iterations {r['iterations']} primal {r['primal']:.6f} dual {r['dual']:.6f} in {time.time() - t0:.1f} s")
    tr = r["trace"]
    for i in range(0, len(tr), 5):
        print(f"  it {i:3d} mu {tr[i,0]:.3e} gap {abs(tr[i,1]-tr[i,2])/(1+abs(tr[i,1])):.2e} pinf {tr[i,3]:.2e} dinf {tr[i,4]:.2e}")
