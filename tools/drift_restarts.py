"""GPU experiment: restarts vs depth on a large drifted topic (objective after a fixed budget)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kafka_assignment_optimizer_amd as kao
kao.init(0)
import importlib.util
spec = importlib.util.spec_from_file_location("hi", os.path.join(os.path.dirname(os.path.abspath(__file__)), "hard_instance.py"))
src = open(spec.origin).read().split("budget = float")[0]
ns = {"__file__": spec.origin}
exec(compile(src, "hi", "exec"), ns)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
t = ns["fam"]["drifted 20 %: B=300, P=2000, 6 racks"]
if len(sys.argv) > 2 and sys.argv[2] == "params":
    for lam_max in (3, 6, 12, 40):
        for period_log2 in (5, 8, 11, 14):
            for obj_scale in (4,):
                r = kao.solve([t], seed=3, restarts=256, iters_per_launch=1024, stop_at_bound=1, time_limit_s=budget, lam_max=lam_max,
                              period_log2=period_log2, obj_scale=obj_scale, dual_iters=-1)[0]
                tm = kao.last_solve_timing()
                print(f"lam_max={lam_max} period_log2={period_log2} S={obj_scale}: {r.status} obj {r.objective} t_best {tm['time_to_best']:.3f} "
                      f"launches {int(tm['launches'])}", flush=True)
    sys.exit(0)
for name in ("drifted 20 %: B=120, P=400, 4 racks", "drifted 20 %: B=300, P=2000, 6 racks"):
    t = ns["fam"][name]
    for restarts in (0, 4096, 1024, 256, 64):
        for ipl in (256, 1024):
            r = kao.solve([t], seed=3, restarts=restarts, iters_per_launch=ipl, stop_at_bound=1, time_limit_s=budget)[0]
            tm = kao.last_solve_timing()
            print(f"{name} restarts={restarts} iters/launch={ipl}: {r.status} obj {r.objective} bound {r.upper_bound} "
                  f"t_best {tm['time_to_best']:.3f} launches {int(tm['launches'])}", flush=True)
