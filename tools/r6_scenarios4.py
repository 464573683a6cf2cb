"""GPU (round 6): the edges of the north star -- heavy drift, half a million partitions, 4,000 brokers, very short limits, the C++ CLI end to
end on a 100,000-partition JSON."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)

def report(name, t, r, dt):
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    ok = "-"
    if r.status not in ("INFEASIBLE_PROVEN", "NO_FEASIBLE"):
        obj, viol = kao.evaluate_batch(t, np.asarray(r.assignment)[None]); ok = f"objective {int(obj[0])} violations {int(np.asarray(viol)[0][0])}"
    print(f"{name}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} read back {tm['results_read_back']:.3f}s (call {dt:.3f}s) launches {tm['launches']} "
          f"lp solves {int(lp['solves'])} iterations {int(lp['iterations'])} adopted {int(lp['adopted'])} fractional {int(lp['fractional_partitions'])} cx {tm['cx_calls']} | evaluator: {ok}", flush=True)

kao.solve([sy.north_star_topic("drift100k")], seed=1, max_launches=1)
cases = [("50 % drift, 1000 x 100,000", lambda: sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], []), 0.5, 1)[0], 3.0),
         ("100 % drift, 1000 x 100,000", lambda: sy.drift(sy.make_cluster(1000, 20, 1, 100_000, 3, [], []), 1.0, 1)[0], 3.0),
         ("1000 x 500,000, 20 % drift", lambda: sy.drift(sy.make_cluster(1000, 20, 1, 500_000, 3, [], []), 0.2, 1)[0], 10.0),
         ("4000 brokers x 100,000, 20 % drift", lambda: sy.drift(sy.make_cluster(4000, 20, 1, 100_000, 3, [], []), 0.2, 1)[0], 10.0)]
for name, mk, lim in cases:
    try:
        t = mk()
        kao.solve([t], seed=1, max_launches=1)
        t0 = time.perf_counter()
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=lim)[0]
        report(f"{name} (limit {lim:g} s)", t, r, time.perf_counter() - t0)
    except Exception as e:
        print(f"{name}: EXCEPTION {e!r}", flush=True)
t = sy.north_star_topic("drift100k")
for lim in (0.02, 0.1, 0.3):
    t0 = time.perf_counter()
    r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=lim)[0]
    report(f"the flagship under time_limit_s = {lim:g}", t, r, time.perf_counter() - t0)
# the C++ CLI end to end
try:
    from kafka_assignment_optimizer_amd.model import assignment_to_json
    cur = {"version": 1, "partitions": [{"topic": "big", "partition": p, "replicas": [int(t.broker_ids[b]) for b in t.current[p]]} for p in range(t.n_partitions)]}
    racks = {str(int(t.broker_ids[b])): f"rack{int(t.rack_of[b])}" for b in range(t.n_brokers)}
    os.makedirs("/tmp/cli", exist_ok=True)
    json.dump(cur, open("/tmp/cli/current.json", "w")); json.dump(racks, open("/tmp/cli/racks.json", "w"))
    blist = ",".join(str(int(b)) for b in t.broker_ids)
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(ROOT, "cli", "kao-cli"), "--current", "/tmp/cli/current.json", "--broker-list", blist, "--racks", "/tmp/cli/racks.json", "--time-limit", "3"],
                       capture_output=True, text=True, timeout=120)
    dt = time.perf_counter() - t0
    out = json.loads(p.stdout) if p.returncode == 0 else None
    print(f"kao-cli on the flagship's JSON ({os.path.getsize('/tmp/cli/current.json') / 1e6:.1f} MB): rc {p.returncode} in {dt:.2f} s, {len(out['partitions']) if out else 0} partitions out; stderr tail: {p.stderr.strip().splitlines()[-1] if p.stderr.strip() else ''}", flush=True)
except Exception as e:
    print(f"kao-cli: EXCEPTION {e!r}", flush=True)
