"""Generates tests/golden/drift_scale.json: exact references for the large drifted single topics of tools/drift_scale.py
(synthetic.drift(make_cluster(B, R, 1, P, 3), 0.2, 1)): the HiGHS MILP optimum where branch-and-bound finishes inside the
time limit, and always the value of the LP relaxation (an upper bound on the optimum; floor(LP) is what a certificate
can reach at best).  "parity unpinned": OUR restatement of the README model (README.md:144-185), not lp_solve output.

Run in the build container:  python tests/golden/make_golden_drift_scale.py B R P [milp_time_limit_s] [lp_method] [drift_seed]
(drift_seed other than 1: the row goes to "rows_other_seeds"; lp_method "none" skips the LP relaxation)
(milp_time_limit_s 0 = the LP relaxation only; lp_method "highs-ipm" for the topics the simplex does not finish: 400 x 3000
took 2,438 s with the interior point method on 8 cores).  Each run merges its row into drift_scale.json.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import kao_oracle as ko  # noqa: E402
from kafka_assignment_optimizer_amd import synthetic  # noqa: E402


def oracle_topic(B, R, P, frac=0.2, seed=1):
    pt = synthetic.drift(synthetic.make_cluster(B, R, 1, P, 3, [], []), frac, seed)[0]
    return ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                    n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights,
                    bounds_override=dict(pt.bounds_override))


def main():
    B, R, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    limit = float(sys.argv[4]) if len(sys.argv) > 4 else 3600.0
    lp_method = sys.argv[5] if len(sys.argv) > 5 else "highs"
    dseed = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    ot = oracle_topic(B, R, P, 0.2, dseed)
    row = {"B": B, "R": R, "P": P, "rf": 3, "drift": 0.2, "seed": dseed,
           "upper_bound_closed_form": int(min(ko.upper_bound_forced(ot), ko.upper_bound_broker(ot)))}
    if lp_method != "none":
        t0 = time.perf_counter()
        lp = ko.lp_bound(ot, lp_method)
        row["lp_value"] = lp
        row["lp_method"] = lp_method
        row["lp_seconds"] = round(time.perf_counter() - t0, 1)
        print("LP", lp, row["lp_seconds"], flush=True)
    if limit > 0:
        ex = ko.solve_exact(ot, limit)
        row["milp_status"] = ex.status
        row["milp_objective"] = ex.objective
        row["milp_seconds"] = round(ex.seconds, 1)
        print("MILP", ex.status, ex.objective, row["milp_seconds"], flush=True)
        if ex.status == "optimal":
            np.save(os.path.join("/tmp", f"drift_{B}_{P}_opt.npy"), ex.assign)
    path = os.path.join(HERE, "drift_scale.json")
    doc = {"rows": []}
    if os.path.exists(path):
        with open(path) as f:
            doc = json.load(f)
    key = "rows" if dseed == 1 else "rows_other_seeds"
    doc[key] = [r for r in doc.get(key, []) if (r["B"], r["R"], r["P"], r.get("seed", 1)) != (B, R, P, dseed)] + [row]
    doc[key].sort(key=lambda r: r["P"])
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
