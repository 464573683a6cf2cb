"""Generates tests/golden/random_wide.json: exact optima (HiGHS on the README model) of the wider random
family oracle/kao_oracle.py::random_case_wide -- uneven racks, RF <= 4 with RF changes, random weights.

Run in the build container (about a minute of HiGHS):  python tests/golden/make_golden_wide.py [N]
"parity unpinned": these are optima of OUR restatement of the README model, not outputs of lp_solve.
"""
import json
import os
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import kao_oracle as ko  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    cases = []
    seed = 0
    t0 = time.time()
    while len(cases) < n:
        tp = ko.random_case_wide(seed)
        seed += 1
        ex = ko.solve_exact(tp, 180)
        if ex.status not in ("optimal", "infeasible"):
            print("seed", seed - 1, "skipped:", ex.status, flush=True)
            continue
        # the instance itself is regenerated from the seed by the test (same generator, oracle/); only a
        # fingerprint of it is stored so that a generator change cannot go unnoticed
        e = {"seed": seed - 1, "shape": [tp.n_brokers, tp.n_racks, tp.n_partitions, tp.rf_cur, tp.rf],
             "current_crc": int(zlib.crc32(tp.current.astype("<u2").tobytes())), "status": ex.status,
             "proven_infeasible_by_counting": bool(ko.provably_infeasible(tp))}
        if ex.status == "optimal":
            e["objective"] = ex.objective
            e["upper_bound"] = min(ko.upper_bound_forced(tp), ko.upper_bound_broker(tp))
        cases.append(e)
        print(len(cases), "seed", seed - 1, "B", tp.n_brokers, "R", tp.n_racks, "P", tp.n_partitions, "rf", tp.rf_cur, "->",
              tp.rf, ex.status, e.get("objective"), f"{time.time() - t0:.0f}s", flush=True)
    with open(os.path.join(HERE, "random_wide.json"), "w") as f:
        json.dump({"cases": cases}, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
