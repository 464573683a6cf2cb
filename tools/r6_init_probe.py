"""Round 6: K-init (one workgroup per restart fills the holes) against the one-wavefront fill inside k_search, config 5 as ONE topic
(1000 brokers x 100,000 partitions, 50 brokers replaced: 15,000 holes).  Prints the first launch's time and checks that every setting
leaves the same restart states.  Run on the GPU box: python tools/r6_init_probe.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic

rng = synthetic.SplitMix64(synthetic.CONFIG_SEED + 5)
rm = rng.sample(list(range(1000)), 50)
add = [(1000 + i, b % 20) for i, b in enumerate(rm)]
pt = synthetic.make_cluster(1000, 20, 1, 100_000, 3, rm, add, bounds_override={"rep_hi": 301})[0]
out = {"workload": "cfg5 as one topic: 1000 x 100,000 RF 3, 15,000 holes", "runs": []}
ref = None
for w in ("0", "1", "4", "8", "16", None):
    if w is None: os.environ.pop("KAO_INIT_WAVES", None)
    else: os.environ["KAO_INIT_WAVES"] = w
    with kao.Session([pt], seed=5, restarts=64, iters_per_launch=16) as s:
        t0 = time.perf_counter(); s.step(1); s.best(); t1 = time.perf_counter()
        st = [s.restart_state(0, rho) for rho in (0, 1, 63)]
    sig = [(d["final"].tobytes(), d["best_obj"], d["V"], d["obj"], d["n_accept"]) for d in st]
    if ref is None: ref = sig
    out["runs"].append({"KAO_INIT_WAVES": w, "first_launch_ms": (t1 - t0) * 1e3, "same_states_as_one_wavefront": sig == ref, "V": st[0]["V"], "obj": st[0]["obj"]})
os.environ.pop("KAO_INIT_WAVES", None)
kao.solve([pt], seed=1, restarts=64, iters_per_launch=16, max_launches=1)
t0 = time.perf_counter()
r = kao.solve([pt], seed=5, restarts=64, iters_per_launch=128, stop_at_bound=1, time_limit_s=1.0)[0]
out["solve"] = {"status": r.status, "objective": int(r.objective), "bound": int(r.upper_bound), "seconds": time.perf_counter() - t0}
print(json.dumps(out))
