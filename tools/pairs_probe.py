"""GPU: the compound-edge layer of KAO-CX (test hook KAO_CX_PAIRS=1) on the committed fixpoint of the drifted 300 x 2000 topic
(14825, one unit below the MILP optimum 14826; the oracle prototype lifts it to 14826) -- test tooling."""
import os, sys, time
os.environ["KAO_CX_PAIRS"] = sys.argv[1] if len(sys.argv) > 1 else "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.drift(sy.make_cluster(300, 6, 1, 2000, 3, [], []), 0.2, 1)[0]
X = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kao_cx_fixpoint_300x2000_d1.npy")).reshape(2000, 3)
t0 = time.time()
Y, obj, st = kao.improve_cycles(t, X, 0)
print(f"KAO_CX_PAIRS={os.environ['KAO_CX_PAIRS']}: objective {st['objective_before']} -> {obj} ({st}) {time.time() - t0:.2f}s", flush=True)
