"""GPU (round 6): two kao_lp_bound calls on a north-star workload (the second is the measured one); what tools/profile_lp_pmc.sh wraps."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "drift100k"
pt = sy.north_star_topic(which)
out = []
for _ in range(2):
    t0 = time.perf_counter(); b = kao.lp_bound(pt); dt = time.perf_counter() - t0
    out.append({"iterations": b["iterations"], "ipm_ms": b["ms"], "call_ms": dt * 1e3, "certificate": b["bound"], "status": b["status"]})
print(json.dumps({"workload": which, "brokers": pt.n_brokers, "racks": pt.n_racks, "partitions": pt.n_partitions, "rf": pt.rf, "runs": out}))
