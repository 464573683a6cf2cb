"""GPU: K-bound alone on one drifted topic -- best dual value after every launch (test tooling).
usage: bound_trace.py B R P target iters launches"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P, target, iters, launches = (int(v) for v in sys.argv[1:7])
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
with kao.Session([t], restarts=4) as s:
    t0 = time.perf_counter()
    for i in range(launches):
        s.bound_step([target], iters)
        b = s.bounds()
        d = s.dual_state(0)
        if i:
            moved = int((d['a'] != prev['a']).sum() + (d['l'] != prev['l']).sum() + (d['g'] != prev['g']).sum())
        else:
            moved = -1
        prev = d
        print(f"launch {i}: {time.perf_counter() - t0:.3f}s iters {b['iters'][0]} flags {b['flags'][0]} bound {b['upper_bound'][0]} best_dual {d['best_dual'] / 65536.0:.3f} multipliers moved {moved}", flush=True)
        if b['flags'][0] & 7:
            break
