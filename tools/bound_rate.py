"""GPU: K-bound iterations per second on large drifted topics, persistent workgroup (KAO_BOUND_CHUNK=0) against the sliced
one-iteration-per-launch kernel with several slice sizes (test tooling).  The rate is the slope between two launch lengths, so
session set-up cancels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
shapes = ((300, 6, 2000), (500, 10, 5000), (500, 10, 10000), (1000, 20, 30000))
chunks = sys.argv[1:] or ["0", "1024", "512", "256", "128"]
for B, R, P in shapes:
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    target = int(kao.upper_bound(t) * 0.99)
    row = []
    for ch in chunks:
        os.environ["KAO_BOUND_CHUNK"] = ch
        n1, n2 = (20, 120) if ch == "0" and P >= 10000 else (100, 600)
        ts = []
        for n in (n1, n2):
            t0 = time.perf_counter()
            got = kao.dual_bound(t, target, iters=n, launches=1)
            ts.append(time.perf_counter() - t0)
        us = (ts[1] - ts[0]) / (n2 - n1) * 1e6
        row.append(f"chunk {ch}: {us:.1f} us/iter (best_dual {got['best_dual']})")
    print(f"B={B} P={P}: " + "; ".join(row), flush=True)
