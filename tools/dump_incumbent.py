"""GPU: solve one drifted topic of tools/drift_scale.py and save the incumbent (test tooling for offline analysis)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
budget = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
dseed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, dseed)[0]
r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]
print(r.status, r.objective, r.upper_bound)
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/inc_{B}_{P}" + (f"_d{dseed}" if dseed != 1 else "") + ".npy", r.assignment)
