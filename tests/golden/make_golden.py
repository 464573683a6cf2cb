"""Generates tests/golden/*.json from the CPU oracle (oracle/kao_oracle.py: HiGHS on the README model).

Run in the build container:  python tests/golden/make_golden.py
The reference snapshot has no tests or fixtures of its own; its single pinned result is the README
worked example (README.md:52-63 -> README.md:85-91), stored as kat1.json.  Everything else here is
"exact optimum of the README model according to HiGHS" -- parity with lp_solve itself is unpinned.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import kao_oracle as ko  # noqa: E402


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", name)


def exact_entry(t, uniq=False):
    ex = ko.solve_exact(t, 300)
    e = {"topic": ko.topic_to_dict(t), "status": ex.status}
    if ex.status == "optimal":
        e["objective"] = ex.objective
        e["assignment"] = ex.assign.tolist()
        e["moves"] = list(ko.count_moves(t, ex.assign))
        e["upper_bound_simple"] = ko.upper_bound_simple(t)
        if uniq:
            e["unique"] = bool(ko.is_unique_optimum(t, ex))
    return e


def main():
    # KAT-1: the README example
    t = ko.readme_example()
    ex = ko.solve_exact(t)
    canon = ko.canonicalize(t, ex.assign)
    rng = np.random.default_rng(1)
    cands = []
    for _ in range(24):
        a = canon.astype(np.int64).copy()
        for _ in range(rng.integers(0, 6)):
            a[rng.integers(0, 10), rng.integers(0, 2)] = rng.integers(0, 21) if rng.random() < 0.9 else 0xFFFF
        obj, viol = ko.verify(t, a)
        cands.append({"assignment": a.tolist(), "objective": obj, "viol": viol.tolist()})
    dump("kat1.json", {
        "source": "README.md:52-63 (current), README.md:43-48 (remove broker 19), README.md:85-91 (expected)",
        "topic": ko.topic_to_dict(t), "objective": ex.objective, "objective_weights_LL4_LF2_FL2_FF1": 49,
        "expected_assignment": canon.tolist(), "expected_moves": [1, 0],
        "expected_json": ko.assignment_to_json([t], [canon]), "n_cooptimal_followers_for_p1": 9,
        "eval_vectors": cands, "lp_text_sha_note": "write_lp(topic) round-trips through solve_lp_text to 58"})

    # BASELINE configs 2-4, first topics (exact optimum by HiGHS)
    for n, k in ((2, 1), (3, 3), (4, 3)):
        case = ko.gen_config(n, n_topics=k)
        dump(f"cfg{n}.json", {"config": n, "removed": case.removed, "added": case.added,
                              "topics": [exact_entry(tp) for tp in case.topics]})

    # small random instances with forced rebalancing, uniqueness flagged
    rnd = []
    seed = 0
    while len(rnd) < 40:
        tp = ko.random_case(seed, max_b=14, max_p=10)
        seed += 1
        if tp.rf > 4 or tp.rf_cur > 4:
            continue
        e = exact_entry(tp, uniq=True)
        e["seed"] = seed - 1
        rnd.append(e)
    dump("random_small.json", {"cases": rnd})

    # medium random instances (up to 40 brokers x 40 partitions): exact optimum, uniqueness flagged
    med = []
    seed = 1000
    while len(med) < 100:
        tp = ko.random_case(seed, max_b=40, max_p=40)
        seed += 1
        if tp.rf > 4 or tp.rf_cur > 4:
            continue
        e = exact_entry(tp, uniq=True)
        if e["status"] not in ("optimal", "infeasible"):
            continue
        e["seed"] = seed - 1
        med.append(e)
    dump("random_medium.json", {"cases": med})


if __name__ == "__main__":
    main()
