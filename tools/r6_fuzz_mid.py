"""GPU fuzz at mid scale (test infrastructure): random drifted clusters of 40-400 brokers x 300-4,000 partitions, RF 2-4, rack-even removals
and additions.  The judge is HiGHS on the COMPACT LP (oracle/kao_lp.py build): a plan must pass the independent verifier and stay at or below
floor(LP); a proven optimum must have objective == certificate <= floor(LP) -- so proven means optimal and the LP had no integrality gap there.
Usage: r6_fuzz_mid.py [n] [first seed]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kao_oracle as ko, kao_lp as kl
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cnt = dict(feasible=0, infeasible=0, proven=0, at_lp=0, wrong=0); secs = []
for i in range(n_cases):
    rng = sy.SplitMix64(0xF1D0000 + s0 + i)
    R = 2 + rng.below(11); per = 4 + rng.below(max(1, 400 // R - 3)); B = R * per
    P = 300 + rng.below(3700); rf = 2 + rng.below(min(3, R + 1 if R < 3 else 3))
    k_rm = rng.below(3) if per > 6 else 0; k_add = rng.below(3)
    rm, add, nid = [], [], B
    for r in range(R):
        rm += rng.sample([b for b in range(B) if b % R == r], k_rm)
        for _ in range(k_add): add.append((nid, r)); nid += 1
    frac = (5 + rng.below(56)) / 100.0
    pt = sy.drift(sy.make_cluster(B, R, 1, P, rf, rm, add), frac, 1 + i)[0]
    ot = ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks, n_partitions=pt.n_partitions, rf=pt.rf,
                  current=np.array(pt.current), weights=pt.weights, bounds_override=dict(pt.bounds_override))
    tag = f"case {s0 + i}: {pt.n_brokers} brokers ({B} - {len(rm)} + {len(add)}) x {P}, {R} racks, RF {rf}, drift {frac:.2f}"
    t0 = time.perf_counter()
    r = kao.solve([pt], seed=3, stop_at_bound=1, time_limit_s=3.0)[0]
    dt = time.perf_counter() - t0
    if r.status in ("INFEASIBLE_PROVEN", "NO_FEASIBLE"):
        cnt["infeasible"] += 1
        if r.status == "NO_FEASIBLE": print(f"{tag}: NO_FEASIBLE after {dt:.2f} s", flush=True)
        continue
    cnt["feasible"] += 1; secs.append(dt)
    val, _, _, _ = kl.solve_highs(kl.build(ot))
    flp = math.floor(val + 1e-6)
    obj, viol = ko.verify(ot, np.asarray(r.assignment))
    bad = None
    if int(np.asarray(viol)[0]) != 0 or obj != r.objective: bad = f"plan fails the verifier ({obj} / {r.objective}, {[int(v) for v in np.asarray(viol)]})"
    elif r.objective > flp: bad = f"objective {r.objective} above floor(LP) {flp}"
    elif r.upper_bound < r.objective: bad = f"certificate {r.upper_bound} below the plan {r.objective}"
    elif r.status == "OPTIMAL_PROVEN" and r.objective != r.upper_bound: bad = "proven with a gap"
    cnt["proven"] += r.status == "OPTIMAL_PROVEN"; cnt["at_lp"] += r.objective == flp
    if bad: cnt["wrong"] += 1
    if bad or r.status != "OPTIMAL_PROVEN":
        print(f"{tag}: {r.status} objective {r.objective} certificate {r.upper_bound} floor(LP) {flp} in {dt:.2f} s{' -- WRONG: ' + bad if bad else ''}", flush=True)
secs.sort()
print(f"{n_cases} instances from seed {s0}: feasible {cnt['feasible']} (proven {cnt['proven']}, objective == floor(LP) on {cnt['at_lp']}), infeasible / none found {cnt['infeasible']}, wrong {cnt['wrong']}; "
      f"solve seconds median {secs[len(secs) // 2] if secs else 0:.2f} max {secs[-1] if secs else 0:.2f}")
