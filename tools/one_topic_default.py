"""kao_solve with DEFAULT options on BASELINE config 5 taken as ONE topic (1000 brokers x 100,000 partitions, RF 3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic
kao.init(0)
rng = synthetic.SplitMix64(synthetic.CONFIG_SEED + 5)
rm = rng.sample(list(range(1000)), 50); add = [(1000 + i, b % 20) for i, b in enumerate(rm)]
pt = synthetic.make_cluster(1000, 20, 1, 100_000, 3, rm, add, bounds_override={"rep_hi": 301})[0]
for rep in range(3):
    t0 = time.perf_counter()
    r = kao.solve([pt], seed=rep, stop_at_bound=1, time_limit_s=5.0)[0]
    dt = time.perf_counter() - t0
    print(rep, r.status, r.objective, r.upper_bound, kao.last_solve_timing(), f"python wall {dt:.3f}s")
t0 = time.perf_counter()
c = kao.canonicalize(pt, r.assignment)
print(f"canonicalize (k_canon, 100k partitions): {time.perf_counter() - t0:.3f}s, changed slots: {int((c != r.assignment).sum())}")
print("still optimal after tie-break:", kao.evaluate(pt, c))
