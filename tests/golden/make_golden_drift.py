"""Generates tests/golden/cfg{2,3,4}_drift.json: HiGHS optima of the first topics of BASELINE configs 2-4 after a 20 % drift
(kafka_assignment_optimizer_amd.synthetic.drift -- pure Python, no GPU needed).  "parity unpinned": optima of OUR
restatement of the README model, not outputs of lp_solve.

Run in the build container:  python tests/golden/make_golden_drift.py [config] [n_topics]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import kao_oracle as ko  # noqa: E402
from kafka_assignment_optimizer_amd import synthetic  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    out = []
    for pt in synthetic.drift(synthetic.make_config(cfg, n_topics=n), 0.2, 1):
        ot = ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                      n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights,
                      bounds_override=dict(pt.bounds_override))
        ex = ko.solve_exact(ot, 600)
        print(pt.name, ex.status, ex.objective, f"{ex.seconds:.1f}s", "closed-form bound", min(ko.upper_bound_forced(ot), ko.upper_bound_broker(ot)), flush=True)
        assert ex.status == "optimal"
        out.append({"topic": ko.topic_to_dict(ot), "status": ex.status, "objective": ex.objective,
                    "upper_bound_closed_form": min(ko.upper_bound_forced(ot), ko.upper_bound_broker(ot))})
    with open(os.path.join(HERE, f"cfg{cfg}_drift.json"), "w") as f:
        json.dump({"config": cfg, "drift": 0.2, "seed": 1, "topics": out}, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
