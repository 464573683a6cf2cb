"""Generates tests/golden/capped_toy.json: exact optima (HiGHS on the joint model, oracle.solve_exact_capped) of small
multi-topic instances under CLUSTER-WIDE per-broker load caps -- every topic's README model (README.md:144-185) plus one
coupling row per capped broker over all topics (BASELINE config 5 "per-broker load caps").
"parity unpinned": OUR restatement, not lp_solve output.

Run in the build container:  python tests/golden/make_golden_capped.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import numpy as np  # noqa: E402
import kao_oracle as ko  # noqa: E402


def toy(seed, n_topics, B0, R, P, rf, removed, added, tight):
    c = ko.make_cluster(f"cap{seed}", B0, R, n_topics, P, rf, removed, added)
    rng = ko._Rng(0xCA9 + seed)
    topics = c.topics
    for t in topics:      # drift a little so that the topics compete for the same brokers
        cur = t.current.copy()
        for _ in range(P // 2):
            p, k, nb = rng.below(P), rng.below(cur.shape[1]), rng.below(t.n_brokers)
            if nb not in cur[p]:
                cur[p, k] = nb
        t.current = cur
    B = topics[0].n_brokers
    lo = sum(t.bounds()["rep_lo"] for t in topics)
    hi = sum(t.bounds()["rep_hi"] for t in topics)
    cap = np.full(B, hi, dtype=int)                          # every topic's own band allows this much in total ...
    for i, b in enumerate(tight):
        cap[b] = lo + (i % 2)                                # ... but a few brokers must stay at (or one above) the floor of EVERY topic
    return topics, cap


def main():
    out = []
    seed = 0
    while len(out) < 6 and seed < 200:
        seed += 1
        rng = ko._Rng(0x5EED0 + seed)
        n_topics = 3 + rng.below(3)
        R = 2 + rng.below(2)
        B0 = R * (3 + rng.below(3))
        P = 5 + rng.below(6)
        rf = 2 + rng.below(2)
        tight = rng.sample(list(range(B0)), 2 + rng.below(3))
        topics, cap = toy(seed, n_topics, B0, R, P, rf, [], [], tight)
        frees = [ko.solve_exact(t) for t in topics]
        if any(f.status != "optimal" for f in frees):
            continue
        free = sum(f.objective for f in frees)
        status, obj, assigns = ko.solve_exact_capped(topics, cap)
        print(seed, status, "capped", obj, "uncapped", free, flush=True)
        if status != "optimal" or obj >= free:
            continue     # keep only instances whose caps bind (the capped optimum is strictly below the sum of the free optima)
        out.append({"seed": seed, "topics": [ko.topic_to_dict(t) for t in topics], "replica_cap": [int(x) for x in cap], "status": status,
                    "objective": obj, "objective_without_caps": free})
    with open(os.path.join(HERE, "capped_toy.json"), "w") as f:
        json.dump({"cases": out}, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
