"""GPU (round 6): KAO-LP's time per iteration against the number of racks at 1000 brokers x 100,000 partitions (the two-level broker rows
and the MFMA rack block need 2 RF + 2 R <= 64 columns per partition; beyond that the round-5 kernels run), and kao_solve under 1 s and 3 s."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
racks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,25,29,30,40,50").split(",")]
for R in racks:
    t = sy.drift(sy.make_cluster(1000, R, 1, 100_000, 3, [], []), 0.2, 1)[0]
    kao.lp_bound(t)
    b = kao.lp_bound(t)
    line = {"racks": R, "iterations": b["iterations"], "ms_per_iteration": b["ms"] / max(1, b["iterations"]), "certificate": b["bound"]}
    kao.solve([t], seed=1, max_launches=1)
    for lim in (1.0, 3.0):
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=lim)[0]
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        line[f"solve_{lim:g}s"] = {"status": r.status, "gap": int(r.upper_bound - r.objective), "seconds": round(tm["results_read_back"], 3), "lp_iterations": int(lp["iterations"])}
    print(json.dumps(line), flush=True)
