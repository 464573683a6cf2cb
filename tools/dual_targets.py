"""GPU: K-bound certificate as a function of the Polyak target (test tooling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P, base = (int(v) for v in sys.argv[1:5])
iters, launches = int(sys.argv[5]), int(sys.argv[6])
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
for d in (int(v) for v in sys.argv[7:]):
    t0 = time.time()
    r = kao.dual_bound(t, base + d, iters, launches)
    print(f"target {base + d}: bound {r['bound']} best_dual {r.get('best_dual')} flags {r.get('flags')} iters {r.get('iters')} {time.time() - t0:.2f}s", flush=True)
