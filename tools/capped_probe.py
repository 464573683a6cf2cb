"""GPU: kao_solve_capped on the capped goldens -- plan total against the exact joint optimum and the Lagrangian bound per case (test tooling).
usage: capped_probe.py [capped_toy.json | capped_medium.json] [time_limit_s]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import kafka_assignment_optimizer_amd as kao
import kao_oracle as ko
from conftest import to_product_topic
kao.init(0)
cases = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "capped_toy.json")))["cases"]
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
eq = 0
for c in cases:
    ots = [ko.topic_from_dict(d) for d in c["topics"]]
    t0 = time.perf_counter()
    res, lb = kao.solve_capped([to_product_topic(t) for t in ots], np.array(c["replica_cap"]), seed=c["seed"], time_limit_s=limit, max_rounds=60)
    dt = time.perf_counter() - t0
    total = sum(int(r.objective) for r in res)
    eq += total == c["objective"]
    print(f"seed {c['seed']}: topics {len(ots)} plan {total} exact {c['objective']} (without caps {c['objective_without_caps']}) lagrangian bound {lb} statuses {sorted(set(r.status for r in res))} {dt:.2f}s", flush=True)
print(f"equal to the exact joint optimum: {eq}/{len(cases)}")
