"""Generates tests/golden/capped_medium.json (round 4; VERDICT r03 item 7: cluster-wide caps beyond toys): ONE cluster of
20 topics x 64 partitions (RF 3) on 60 brokers / 4 racks, every topic drifted, a dozen brokers capped at (or one above) the
sum of the topics' floors -- exact joint optimum by HiGHS (oracle.solve_exact_capped: every topic's README rows,
README.md:144-185, plus one coupling row per capped broker over all topics; BASELINE config 5 "per-broker load caps").
"parity unpinned": OUR restatement, not lp_solve output.

Run in the build container:  python tests/golden/make_golden_capped_medium.py   (minutes)
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, HERE)
import kao_oracle as ko  # noqa: E402
from make_golden_capped import toy  # noqa: E402


def main():
    out = []
    for seed, (n_topics, B0, R, P, n_tight) in ((101, (20, 60, 4, 64, 12)), (102, (12, 40, 4, 48, 8))):
        rng = ko._Rng(0x5EED0 + seed)
        tight = rng.sample(list(range(B0)), n_tight)
        topics, cap = toy(seed, n_topics, B0, R, P, 3, [], [], tight)
        t0 = time.time()
        frees = [ko.solve_exact(t) for t in topics]
        assert all(f.status == "optimal" for f in frees)
        free = sum(f.objective for f in frees)
        status, obj, assigns = ko.solve_exact_capped(topics, cap, time_limit=3000.0)
        print(seed, status, "capped", obj, "uncapped", free, f"{time.time() - t0:.0f}s", flush=True)
        if status != "optimal":
            continue
        out.append({"seed": seed, "topics": [ko.topic_to_dict(t) for t in topics], "replica_cap": [int(x) for x in cap], "status": status,
                    "objective": obj, "objective_without_caps": free})
    with open(os.path.join(HERE, "capped_medium.json"), "w") as f:
        json.dump({"cases": out}, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
