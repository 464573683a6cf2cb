"""GPU: kao_solve on the 400-instance wide golden family in one call -- time, objective parity, proofs (test tooling)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kao_oracle as ko
import kafka_assignment_optimizer_amd as kao
from conftest import to_product_topic
kao.init(0)
cases = json.load(open(os.path.join(ROOT, "tests/golden/random_wide.json")))["cases"]
pts = [to_product_topic(ko.random_case_wide(c["seed"])) for c in cases]
for dual in ((0,) if os.environ.get('KAO_WIDE_DUAL_ONLY') else (-1, 0)):
    for rep in range(2):
        t0 = time.perf_counter()
        res = kao.solve(pts, seed=31, restarts=32, iters_per_launch=256, time_limit_s=5.0, stop_at_bound=1, dual_iters=dual)
        dt = time.perf_counter() - t0
    feas = [(c, r) for c, r in zip(cases, res) if c["status"] == "optimal"]
    print(f"dual_iters={dual}: {dt*1e3:.1f} ms wall, timing {kao.last_solve_timing()}, optimal objective on "
          f"{sum(r.objective == c['objective'] for c, r in feas)}/{len(feas)}, proven {sum(r.status == 'OPTIMAL_PROVEN' for c, r in feas)}, "
          f"statuses { {s: sum(r.status == s for r in res) for s in set(r.status for r in res)} }")
