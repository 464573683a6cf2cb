"""GPU: KAO-CX alone on a saved incumbent (an .npy assignment, e.g. extracted from `r3_probe.py dump`) -- test tooling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P, rounds = (int(v) for v in sys.argv[1:5])
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
A = np.load(sys.argv[5])
t0 = time.time()
X, obj, st = kao.improve_cycles(t, A, rounds)
print(B, P, st, "%.3f s" % (time.time() - t0))
