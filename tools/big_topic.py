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
    return sy.north_star_topic(which)


def steps(which, launches=6, iters=512, restarts=0):
    return sy.north_star_steps(kao, which, launches, iters, restarts)


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
        extra = {"restarts": int(os.environ["RESTARTS"])} if os.environ.get("RESTARTS") else {}
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget, **extra)[0]
        tm = kao.last_solve_timing()
        print(json.dumps({"workload": which, "status": str(r.status), "objective": int(r.objective), "certificate": int(r.upper_bound),
                          "seconds": time.perf_counter() - t0, "timing": tm}))
