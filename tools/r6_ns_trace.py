import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["KAO_SOLVE_TRACE"] = os.environ.get("KAO_SOLVE_TRACE", "1")
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.north_star_topic("drift100k")
kao.solve([t], seed=1, max_launches=1)
for rep in range(int(os.environ.get('REPS', '2'))):
    t0 = time.perf_counter()
    r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=1.0)[0]
    dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print("RESULT", r.status, r.objective, r.upper_bound, "call %.3f" % dt, tm, lp, flush=True)
