"""GPU fuzz sweep (test infrastructure, not product): random instance shapes -> device K-search / K-eval vs the
scalar replay (oracle/kao_port.c) and the independent numpy verifier (oracle/kao_oracle.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kao_oracle as ko, kao_port as kp
import kafka_assignment_optimizer_amd as kao
from conftest import to_product_topic, random_candidates

kao.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big_every = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # every n-th topic is large
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dual_iters = int(sys.argv[5]) if len(sys.argv) > 5 else 25   # K-bound iterations per launch (2 launches)
rng = ko._Rng(0xF022 + seed0)
topics = []
while len(topics) < n_cases:
    R = 1 + rng.below(12)
    B0 = max(R + 1, 3 + rng.below(180))
    P = 1 + rng.below(300)
    if big_every and len(topics) % big_every == big_every - 1:
        P = 2500 + rng.below(6000)  # large topic: 2 or 1 restarts per workgroup, or the global-memory path
    rf = 1 + rng.below(min(4, B0 - 1))
    n_rm = rng.below(max(1, B0 // 4))
    n_add = rng.below(1 + B0 // 8)
    rm = rng.sample(list(range(B0)), n_rm)
    add = [(B0 + i, rng.below(R)) for i in range(n_add)]
    new_rf = rf
    if rng.below(3) == 0:
        new_rf = max(1, min(4, rf + (1 if rng.below(2) else -1), B0 - n_rm + n_add - 1))
    if B0 - n_rm + n_add < max(new_rf, 1) + 0:
        continue
    try:
        t = ko.make_cluster(f"f{len(topics)}", B0, R, 1, P, rf, rm, add, new_rf=new_rf).topics[0]
    except Exception:
        continue
    if t.n_brokers < t.rf:
        continue
    if rng.below(4) == 0:  # random objective weights (leader weight stays the largest)
        lf, fl, ff = 1 + rng.below(3), 1 + rng.below(3), 1 + rng.below(3)
        t.weights = ((max(lf, fl, ff) + 1 + rng.below(3), lf), (fl, ff))
    if rng.below(4) == 0:  # random band overrides (caps / floors), kept satisfiable on average
        bd = t.bounds()
        t.bounds_override = {"rep_hi": bd["rep_hi"] + rng.below(3), "lead_hi": bd["lead_hi"] + rng.below(2)}
        if rng.below(2):
            t.bounds_override["rack_hi"] = bd["rack_hi"] + rng.below(4)
            t.bounds_override["rack_lo"] = max(0, bd["rack_lo"] - rng.below(4))
        if rng.below(3) == 0:
            t.bounds_override["prack_hi"] = bd["prack_hi"] + 1
    if rng.below(3) == 0:  # scramble part of the start
        cur = t.current.copy()
        for _ in range(rng.below(P + 1)):
            p = rng.below(P); k = rng.below(cur.shape[1]); nb = rng.below(t.n_brokers)
            if nb not in cur[p]:
                cur[p, k] = nb
        t.current = cur
    topics.append(t)
print(f"{len(topics)} topics; B range {min(t.n_brokers for t in topics)}..{max(t.n_brokers for t in topics)}, "
      f"P range {min(t.n_partitions for t in topics)}..{max(t.n_partitions for t in topics)}")
bad = 0
t0 = time.time()
for lo in range(0, len(topics), 50):
    batch = topics[lo:lo + 50]
    pts = [to_product_topic(t) for t in batch]
    seed = 1000 + lo
    with kao.Session(pts, seed=seed, restarts=8, iters_per_launch=80) as s:
        s.step(launches)
        if s.stats()["drift"] != 0:
            bad += 1; print("DRIFT", lo)
        for ti, ot in enumerate(batch):
            tseed = seed ^ (((ti + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
            rho = (ti * 3) % 8
            dev = s.restart_state(ti, rho)
            ref = kp.port_search(ot, tseed, rho, launches, 80)
            if dev["final"].tolist() != ref["final"].tolist() or (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) != (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"]):
                bad += 1; print("REPLAY MISMATCH", lo + ti, ot.n_brokers, ot.n_racks, ot.n_partitions, ot.rf_cur, ot.rf)
            obj, viol = ko.verify(ot, dev["final"])
            if (obj, int(viol[0])) != (dev["obj"], dev["V"]):
                bad += 1; print("VERIFIER MISMATCH", lo + ti, (obj, int(viol[0])), (dev["obj"], dev["V"]))
        for ot, r in zip(batch, s.best()):
            if r.status not in ("NO_FEASIBLE", "INFEASIBLE_PROVEN"):
                obj, viol = ko.verify(ot, r.assignment)
                if viol[0] != 0 or obj != r.objective or r.objective > r.upper_bound:
                    bad += 1; print("BEST MISMATCH", ot.name, obj, viol.tolist(), r.objective, r.upper_bound)
    for ti, (ot, pt) in enumerate(zip(batch, pts)):  # K-bound vs its replay: multipliers and dual value, two launches
        if ti % 3 or ko.provably_infeasible(ot) or ot.n_brokers > 8192 or max(max(w) for w in ot.weights) > 255:
            continue
        target = max(0, ko.upper_bound_simple(ot) - 3 - (ti % 5))
        try:
            dv = kao.dual_bound(pt, target, iters=dual_iters, launches=2)
        except kao.KaoError as e:
            bad += 1; print("DUAL ERROR", ot.name, e); continue
        st = kp.DualState(ot)
        for _ in range(2):
            st = kp.port_dual_bound(ot, target, dual_iters, st)
            if st.flags & 7:
                break
        if (dv["iters"], dv["flags"], dv["best_dual"] if dv["iters"] else 0) != (st.iters, st.flags, st.best_L if st.iters else 0) or \
                dv["a"].tolist() != st.a.tolist() or dv["l"].tolist() != st.l.tolist() or dv["g"].tolist() != st.g[:ot.n_racks].tolist():
            bad += 1; print("DUAL MISMATCH", lo + ti, ot.n_brokers, ot.n_racks, ot.n_partitions, ot.rf_cur, ot.rf, dv["iters"], st.iters, dv["flags"], st.flags)
    for ot, pt in list(zip(batch, pts))[:10]:  # K-eval on mutated candidates
        cands = random_candidates(ot, 5, seed=lo, p_mut=0.3, p_none=0.05)
        o, v = kao.evaluate_batch(pt, cands)
        for i in range(len(cands)):
            oo, vv = ko.verify(ot, cands[i])
            if (int(o[i]), v[i].tolist()) != (oo, vv.tolist()):
                bad += 1; print("EVAL MISMATCH", ot.name, i)
print(f"fuzz done: {len(topics)} topics, {bad} problems, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
