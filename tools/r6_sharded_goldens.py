"""GPU (round 6): the sharded LP (kao_lp_sharded_test, logical shards) against the whole-topic LP on the golden families -- KAT-1, RF 5..8
(C5 rows and bounded C7 slacks live), the medium family (RF = R: dependent local rows), a topic with broker weights: same certificate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["KAO_RCCL_LOOPBACK"] = "1"
import json
import numpy as np
import kafka_assignment_optimizer_amd as kao
import kao_oracle as ko
from conftest import load_golden, to_product_topic
kao.init(0)
cases = [("KAT-1", ko.readme_example(), 58)]
for c in load_golden("random_rf.json")["cases"][:40]:
    if c["status"] == "optimal": cases.append(("rf seed %d" % c["seed"], ko.random_case_rf(c["seed"]), c["objective"]))
for c in load_golden("random_medium.json")["cases"][:30]:
    if c["status"] == "optimal": cases.append(("medium seed %d" % c["seed"], ko.topic_from_dict(c["topic"]), c["objective"]))
n = same = 0
for name, ot, opt in cases:
    pt = to_product_topic(ot)
    whole = kao.lp_bound(pt)
    for shards in (2, 3):
        if pt.n_partitions < shards: continue
        sh = kao.lp_sharded(pt, [0] * shards, pert=-1.0, tol=1e-7, max_iters=80)
        n += 1; ok = sh["bound"] == whole["bound"]; same += ok
        if not ok or n <= 3: print(f"{name} ({pt.n_brokers} brokers, {pt.n_partitions} partitions, rf {pt.rf}) in {shards} shards: certificate {sh['bound']} (whole {whole['bound']}, optimum {opt}), {sh['iterations']} / {whole['iterations']} iterations, status {sh['status']} / {whole['status']}", flush=True)
print(f"sharded against whole: {same} of {n} certificates equal")
