"""GPU: K-bound ALONE on one drifted topic (what tools/profile_bound.sh wraps in rocprofv3): `launches` launches of `iters` iterations.
usage: bound_only.py B R P [iters] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 4
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
target = int(kao.upper_bound(t) * 0.9)     # far below: no launch stops early
t0 = time.perf_counter()
got = kao.dual_bound(t, target, iters=iters, launches=launches)
dt = time.perf_counter() - t0
print(f"K-bound alone B={B} P={P}: {got['iters']} iterations in {dt:.3f} s (incl. session set-up) = {dt / max(got['iters'], 1) * 1e6:.1f} us/iteration, "
      f"dual {got['best_dual']} flags {got['flags']}")
