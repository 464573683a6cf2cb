"""GPU: the north-star regime as a workload (test / profiling tooling; bench.py's roofline_big_topic leg runs the same steps).
  big_topic.py steps <which> [launches]   K-search + K-eval launches of a session (what rocprofv3 --pmc wraps: equal launches)
  big_topic.py solve <which> [seconds]    one kao_solve call (K-search, K-eval, K-bound, KAO-CX: where the GPU time goes)
which: drift30k = drifted 1000 brokers x 30,000 partitions (RF 3, 20 racks); cfg5one = BASELINE config 5 as ONE topic
(1000 brokers, 100,000 partitions, 50 brokers replaced, per-broker cap ceil(avg)+1); drift5k = drifted 500 x 5,000."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy


def instance(which):
    if which == "drift30k":
        return sy.drift(sy.make_cluster(1000, 20, 1, 30000, 3, [], []), 0.2, 1)[0]
    if which == "drift5k":
        return sy.drift(sy.make_cluster(500, 10, 1, 5000, 3, [], []), 0.2, 1)[0]
    if which == "cfg5one":
        rng = sy.SplitMix64(sy.CONFIG_SEED + 5)
        rm = rng.sample(list(range(1000)), 50)
        add = [(1000 + i, b % 20) for i, b in enumerate(rm)]
        return sy.make_cluster(1000, 20, 1, 100_000, 3, rm, add, bounds_override={"rep_hi": 301})[0]
    raise SystemExit("which: drift30k | drift5k | cfg5one")


def steps(which, launches=6, iters=512, restarts=0):
    """`launches` K-search + K-eval steps after the init launch; returns the per-launch figures of the timed ones."""
    t = instance(which)
    opts = dict(seed=3, iters_per_launch=iters, profile=1)
    if restarts:
        opts["restarts"] = restarts
    with kao.Session([t], **opts) as s:
        s.step(1)                      # launch 0: best-insertion init + the first iterations (not timed)
        s.sync()
        a = s.stats()
        t0 = time.perf_counter()
        s.step(launches)
        s.sync()
        wall = time.perf_counter() - t0
        b = s.stats()
        best = s.best()[0]
    n = launches
    rf, P, B = t.rf, t.n_partitions, t.n_brokers
    nb = (b["delta_candidates"] - a["delta_candidates"]) / n
    return {"workload": which, "brokers": B, "partitions": P, "rf": rf, "restarts": b["n_restarts_total"], "iters_per_launch": iters,
            "launches_timed": n, "k_search_ms_per_launch": (b["ms_search"] - a["ms_search"]) / n, "k_eval_ms_per_launch": (b["ms_eval"] - a["ms_eval"]) / n,
            "wall_ms_per_launch": 1e3 * wall / n, "neighbours_per_launch": nb, "k_search_algorithmic_bytes_per_launch": nb * (8 * rf + 10),
            "k_eval_algorithmic_bytes_per_launch": b["n_restarts_total"] * (4 * rf * P + B), "k_search_lds_bytes": b["lds_bytes_search"],
            "k_search_workgroups": b["blocks_search"], "objective_after": int(best.objective), "violation_after": int(best.violations[0]), "drift": b["drift"]}


if __name__ == "__main__":
    kao.init(0)
    mode, which = sys.argv[1], sys.argv[2]
    if mode == "steps":
        print(json.dumps(steps(which, int(sys.argv[3]) if len(sys.argv) > 3 else 6)))
    else:
        t = instance(which)
        budget = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
        kao.solve([t], seed=1, max_launches=1)          # arenas, code objects
        t0 = time.perf_counter()
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]
        tm = kao.last_solve_timing()
        print(json.dumps({"workload": which, "status": str(r.status), "objective": int(r.objective), "certificate": int(r.upper_bound),
                          "seconds": time.perf_counter() - t0, "timing": tm}))
