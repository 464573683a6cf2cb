"""GPU probe (test tooling): the actual margins behind the tolerant assertions of tests/test_gpu_parity.py / test_gpu_cycle.py, so
that the thresholds can be set to what the device delivers.  One line per measurement."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
import kao_oracle as ko
from conftest import to_product_topic, load_golden

kao.init(0)
what = set((sys.argv[1] if len(sys.argv) > 1 else "slack,other,scale,medium,wide").split(","))


def topic(B, R, P, dseed=1):
    return sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, dseed)[0]


if "slack" in what:
    rows = {(r["B"], r["R"], r["P"]): r for r in load_golden("drift_scale.json")["rows"]}
    for shape in ((270, 6, 2200), (350, 7, 2500), (450, 9, 3500)):
        lp = int(round(rows[shape]["lp_value"]))
        for seed in (1, 2, 3):
            r = kao.solve([topic(*shape)], seed=seed, time_limit_s=4.0)[0]
            print(f"slack {shape} seed {seed}: {r.status} objective {r.objective} certificate {r.upper_bound} LP {lp}", flush=True)
if "other" in what:
    other = load_golden("drift_scale.json")["rows_other_seeds"][0]
    t2 = topic(other["B"], other["R"], other["P"], other["seed"])
    for seed in (1, 2, 3, 4, 5):
        for lim in (4.0, 8.0):
            r = kao.solve([t2], seed=seed, time_limit_s=lim)[0]
            print(f"other-seed 300x2000 d2 seed {seed} limit {lim}: {r.status} objective {r.objective} certificate {r.upper_bound} MILP {other['milp_objective']}", flush=True)
if "scale" in what:
    rows = {(r["B"], r["R"], r["P"]): r for r in load_golden("drift_scale.json")["rows"]}
    for shape in ((400, 8, 3000), (250, 5, 4000)):
        for seed in (1, 2, 3):
            r = kao.solve([topic(*shape)], seed=seed, time_limit_s=6.0)[0]
            print(f"scale {shape} seed {seed}: {r.status} objective {r.objective} certificate {r.upper_bound} LP {rows[shape]['lp_value']}", flush=True)
if "medium" in what:
    cases = load_golden("random_medium.json")["cases"]
    ots = [ko.topic_from_dict(c["topic"]) for c in cases]
    res = kao.solve([to_product_topic(t) for t in ots], seed=23, restarts=64, iters_per_launch=512, max_launches=10, time_limit_s=60.0)
    n_opt = sum(c["status"] != "infeasible" for c in cases)
    n_eq = sum(c["status"] != "infeasible" and r.objective == c["objective"] for c, r in zip(cases, res))
    n_pr = sum(c["status"] != "infeasible" and r.status == "OPTIMAL_PROVEN" for c, r in zip(cases, res))
    print(f"random_medium: n_opt {n_opt} equal {n_eq} proven {n_pr}", flush=True)
if "wide" in what:
    os.environ["KAO_CX_EAGER"] = "1"
    cases = load_golden("random_wide.json")["cases"]
    ots = [ko.random_case_wide(c["seed"]) for c in cases]
    res = kao.solve([to_product_topic(t) for t in ots], seed=31, restarts=32, iters_per_launch=256, time_limit_s=20.0, stop_at_bound=1)
    n_opt = sum(c["status"] == "optimal" for c in cases)
    n_eq = sum(c["status"] == "optimal" and r.objective == c["objective"] for c, r in zip(cases, res))
    n_pr = sum(c["status"] == "optimal" and r.status == "OPTIMAL_PROVEN" for c, r in zip(cases, res))
    print(f"random_wide with eager KAO-CX: n_opt {n_opt} equal {n_eq} proven {n_pr}", flush=True)
