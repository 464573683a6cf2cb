"""Wall-clock breakdown of a whole solve (config 4 / 5): where time-to-optimal goes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic
from kafka_assignment_optimizer_amd.solver import decode_key
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 128
kao.init(0)
topics = synthetic.make_config(cfg)
ub = [kao.upper_bound(t) for t in topics]
for rep in range(3):
    t0 = time.perf_counter()
    s = kao.Session(topics, seed=5 + rep, iters_per_launch=iters)
    t1 = time.perf_counter()
    n = 0
    while True:
        s.step(1); keys = s.best_keys(); n += 1
        done = all(decode_key(k)[0] == 0 and decode_key(k)[1] >= u for k, u in zip(keys, ub))
        if done or n >= 50: break
    t2 = time.perf_counter()
    res = s.best()
    t3 = time.perf_counter()
    s.close()
    t4 = time.perf_counter()
    tt = time.perf_counter()
    r2 = kao.solve(topics, seed=5 + rep, iters_per_launch=iters, stop_at_bound=1, time_limit_s=20)
    t5 = time.perf_counter()
    tm = kao.last_solve_timing()
    print("   C timing ms:", {k: (round(1e3*v,3) if k != "launches" else v) for k, v in tm.items()})
    print(f"cfg{cfg} rep{rep}: create {1e3*(t1-t0):.2f} ms | {n} launches+polls {1e3*(t2-t1):.2f} ms | best {1e3*(t3-t2):.2f} ms | close {1e3*(t4-t3):.2f} ms | kao_solve total {1e3*(t5-tt):.2f} ms, all proven: {all(r.status=='OPTIMAL_PROVEN' for r in r2)}")
