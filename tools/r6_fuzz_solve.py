"""GPU fuzz of kao_solve against the exact solver of the README model (HiGHS through oracle/kao_oracle.py; test infrastructure): fresh seeds
of the wide and the RF families at larger sizes than the committed goldens.  A proven optimum must be the exact optimum, a certificate must
not undercut it, an infeasible instance must not get a plan, a feasible one must not be called infeasible.
Usage: r6_fuzz_solve.py [n_wide] [n_rf] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kao_oracle as ko
import kafka_assignment_optimizer_amd as kao
from conftest import to_product_topic
kao.init(0)
n_wide = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n_rf = int(sys.argv[2]) if len(sys.argv) > 2 else 60
s0 = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
cases = [("wide", s, ko.random_case_wide(s, max_b=90, max_p=220)) for s in range(s0, s0 + n_wide)] + [("rf", s, ko.random_case_rf(s, max_b=40, max_p=60)) for s in range(s0, s0 + n_rf)]
n = dict(optimal=0, infeasible=0, skipped=0, proven=0, found=0, wrong=0)
t_begin = time.perf_counter()
for fam, seed, ot in cases:
    ex = ko.solve_exact(ot, time_limit=20.0)
    if ex.status not in ("optimal", "infeasible"):
        n["skipped"] += 1; continue
    pt = to_product_topic(ot)
    r = kao.solve([pt], seed=3, stop_at_bound=1, time_limit_s=2.0)[0]
    bad = None
    if ex.status == "infeasible":
        n["infeasible"] += 1
        if r.status not in ("INFEASIBLE_PROVEN", "NO_FEASIBLE"): bad = f"a plan ({r.status}, {r.objective}) for an infeasible instance"
    else:
        n["optimal"] += 1
        if r.status in ("INFEASIBLE_PROVEN",): bad = "called infeasible"
        elif r.status != "NO_FEASIBLE":
            obj, viol = ko.verify(ot, np.asarray(r.assignment))
            if int(np.asarray(viol)[0]) != 0 or obj != r.objective: bad = f"plan fails the verifier: {obj} / {r.objective}, violations {[int(v) for v in np.asarray(viol)]}"
            elif r.objective > ex.objective: bad = f"objective {r.objective} above the exact optimum {ex.objective}"
            elif r.upper_bound < ex.objective: bad = f"certificate {r.upper_bound} below the exact optimum {ex.objective}"
            elif r.status == "OPTIMAL_PROVEN" and r.objective != ex.objective: bad = f"proven {r.objective}, exact {ex.objective}"
            n["proven"] += r.status == "OPTIMAL_PROVEN"; n["found"] += r.objective == ex.objective
    if bad:
        n["wrong"] += 1
        print(f"WRONG {fam} seed {seed} (B {ot.n_brokers} R {ot.n_racks} P {ot.n_partitions} RF {ot.rf}): {bad}", flush=True)
print(f"{len(cases)} instances from seed {s0}: exact solver optimal {n['optimal']}, infeasible {n['infeasible']}, skipped (its 20-s limit) {n['skipped']}; kao_solve found the optimum on {n['found']}, "
      f"proved it on {n['proven']} of {n['optimal']}; wrong answers {n['wrong']}; {time.perf_counter() - t_begin:.0f} s")
