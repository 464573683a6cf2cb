"""GPU (round 6): the perturbed interior-point solve on two racks (600 brokers x 50,000 partitions ran into the 200-iteration cap in
tools/r6_scenarios3.py while 60 x 2,000 takes 24): the trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KAO_LP_TRACE_PERT"] = "-1"
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for (B, R, P) in ((600, 2, 50_000), (600, 2, 10_000), (600, 4, 50_000)):
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    kao.lp_trace(t, max_iters=1)
    d = kao.lp_trace(t, tol=1e-10, max_iters=200)
    print(f"{B} x {P}, {R} racks: status {d['status']} iterations {d['iterations']} {d['ms']:.0f} ms")
    for i, row in enumerate(d["trace"]):
        if i % 10 == 0 or i >= len(d["trace"]) - 2:
            print(f"  it {i:3d} mu {row[0]:.3e} pobj {row[1]:.6f} dobj {row[2]:.6f} gap {abs(row[1]-row[2])/(1+abs(row[1])):.2e} pinf {row[3]:.2e} dinf {row[4]:.2e}")
