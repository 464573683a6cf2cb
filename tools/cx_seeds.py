"""GPU: kao_solve on one drifted topic over several seeds; mean objective / certificate (test tooling)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
B, R, P = (int(v) for v in sys.argv[1:4])
budget = float(sys.argv[4]); n = int(sys.argv[5])
t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
kw = {k: int(v) for k, v in (a.split('=') for a in os.environ.get('KW', '').split(',') if a)}
objs, ubs, proven = [], [], 0
for seed in range(3, 3 + n):
    r = kao.solve([t], seed=seed, stop_at_bound=1, time_limit_s=budget, **kw)[0]
    objs.append(r.objective); ubs.append(r.upper_bound); proven += r.status == "OPTIMAL_PROVEN"
print(f"{B}x{P} {os.environ.get('KW', '')}: objectives {objs} mean {sum(objs) / n:.2f} certificates {ubs} proven {proven}/{n}", flush=True)
