"""Generates tests/golden/random_rf.json: HiGHS optima of the high-replication-factor family oracle.random_case_rf (5..8
replicas, RF changes across the 4 / 5 boundary): README.md:148-151 puts no cap on the replication factor.  "parity unpinned":
optima of OUR restatement of the README model, not outputs of lp_solve.

Run in the build container:  python tests/golden/make_golden_rf.py [n_cases]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import kao_oracle as ko  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    cases = []
    for seed in range(n):
        t = ko.random_case_rf(seed)
        ex = ko.solve_exact(t, 300)
        if ex.status not in ("optimal", "infeasible"):
            continue
        e = {"seed": seed, "status": ex.status, "rf_cur": t.rf_cur, "rf": t.rf, "B": t.n_brokers, "P": t.n_partitions}
        if ex.status == "optimal":
            e["objective"] = ex.objective
            e["assignment"] = ex.assign.tolist()
            e["upper_bound"] = int(min(ko.upper_bound_forced(t), ko.upper_bound_broker(t)))
            e["unique"] = bool(ko.is_unique_optimum(t, ex)) if t.n_brokers * t.n_partitions <= 300 else False
        cases.append(e)
        print(e["seed"], e["status"], e.get("objective"), flush=True)
    with open(os.path.join(HERE, "random_rf.json"), "w") as f:
        json.dump({"generator": "kao_oracle.random_case_rf(seed)", "cases": cases}, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
