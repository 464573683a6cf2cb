"""GPU: K-bound alone on one drifted topic -- best dual value along a schedule of targets (test tooling).
usage: bound_trace.py B R P iters_per_launch upto:target [upto:target ...]   (iteration counts are cumulative)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P, iters = (int(v) for v in sys.argv[1:5])
sched = [tuple(int(x) for x in a.split(":")) for a in sys.argv[5:]]
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
with kao.Session([t], restarts=4) as s:
    t0 = time.perf_counter()
    done, prev = 0, None
    for upto, target in sched:
        while done < upto:
            s.bound_step([target], iters)
            done += iters
        b = s.bounds()
        d = s.dual_state(0)
        moved = -1 if prev is None else int((d['a'] != prev['a']).sum() + (d['l'] != prev['l']).sum() + (d['g'] != prev['g']).sum())
        prev = d
        print(f"{time.perf_counter() - t0:.3f}s iters {b['iters'][0]} target {target} flags {b['flags'][0]} bound {b['upper_bound'][0]} "
              f"best_dual {d['best_dual'] / 65536.0:.3f} multipliers moved {moved}", flush=True)
        if b['flags'][0] & 7:
            break
