"""GPU (round 6): per-iterate trace of the PERTURBED interior-point solve (what kao_solve runs on a huge topic) on drift seeds of the
north-star size: where the iterations go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KAO_LP_TRACE_PERT"] = "-1"
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for ds in [int(x) for x in (sys.argv[1:] or ["1", "3"])]:
    t = sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], []), 0.2, ds)[0]
    kao.lp_trace(t, max_iters=1)
    d = kao.lp_trace(t, tol=1e-10, max_iters=200)
    print(f"drift seed {ds}: status {d['status']} iterations {d['iterations']} {d['ms']:.0f} ms")
    for i, row in enumerate(d["trace"]):
        if i % 5 == 0 or i >= len(d["trace"]) - 3:
            print(f"  it {i:3d} mu {row[0]:.3e} pobj {row[1]:.6f} dobj {row[2]:.6f} gap {abs(row[1]-row[2])/(1+abs(row[1])):.2e} pinf {row[3]:.2e} dinf {row[4]:.2e}")
