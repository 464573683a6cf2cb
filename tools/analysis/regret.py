"""CPU (round 5): where does a KAO-CX fixpoint differ from an optimum?  With the LP's optimal multipliers (a, l, g) the Lagrangian
splits by partition; the REGRET of a partition = (best priced value of any of its configurations) - (priced value of the one the
incumbent holds) >= 0, and  sum of regrets + complementary-slackness loss of the band rows = certificate - incumbent  (up to the
LP's tolerance).  So the partitions with a positive regret, and the brokers / racks whose band row is priced but not at the priced
end, are exactly where an improving move has to act.
Usage: regret.py  (the committed 300 x 2000 fixpoint, 14825 against the optimum 14826)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kao_oracle as ko
import kao_lp as klp


def drifted(B, R, P, seed=1, frac=0.2):
    from helpers import drift_case   # tests/helpers.py
    return drift_case(ko, B, R, P, seed, frac)


def analyse(t, X, a, l, g, S=65536.0):
    a = np.asarray(a, float) / S; l = np.asarray(l, float) / S; g = np.asarray(g, float) / S
    B, R, P, RF = t.n_brokers, t.n_racks, t.n_partitions, t.rf
    rack = np.asarray(t.rack_of)
    w = t.weights   # [[w00, w01], [w10, w11]]: [current role][new role], role 0 = leader
    cur = t.current
    regrets = np.zeros(P); best_cfg = [None] * P
    prack_lo, prack_hi = ko._floor_ceil(RF, R)
    assert prack_hi == 1, "analysis written for one replica per rack at most"
    base_f = -a - g[rack]; base_l = -a - g[rack] - l
    for p in range(P):
        vf = base_f.copy(); vl = base_l.copy()
        c = [int(b) for b in cur[p] if b >= 0]
        for j, b in enumerate(c):
            vl[b] += w[0][0] if j == 0 else w[1][0]
            vf[b] += w[0][1] if j == 0 else w[1][1]
        # best follower / leader per rack
        Fr = np.full(R, -1e18); Fb = np.zeros(R, int); Lr = np.full(R, -1e18); Lb = np.zeros(R, int)
        for r in range(R):
            m = np.nonzero(rack == r)[0]
            i = m[np.argmax(vf[m])]; Fr[r] = vf[i]; Fb[r] = i
            i = m[np.argmax(vl[m])]; Lr[r] = vl[i]; Lb[r] = i
        best = -1e18; cfg = None
        for r0 in range(R):
            others = sorted((r for r in range(R) if r != r0), key=lambda r: -Fr[r])[:RF - 1]
            v = Lr[r0] + sum(Fr[r] for r in others)
            if v > best: best = v; cfg = (int(Lb[r0]),) + tuple(int(Fb[r]) for r in others)
        x = [int(b) for b in X[p]]
        val = vl[x[0]] + sum(vf[b] for b in x[1:])
        regrets[p] = best - val; best_cfg[p] = cfg
    return regrets, best_cfg


def topic(B, R, P, rf=3, seed=1):
    from kafka_assignment_optimizer_amd import synthetic as sy
    pt = sy.drift(sy.make_cluster(B, R, 1, P, rf, [], []), 0.2, seed)[0]
    return ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                    n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights)


if __name__ == "__main__":
    B, R, P = 300, 6, 2000
    t = topic(B, R, P)
    X = np.load(os.path.join(ROOT, "tests", "golden", "kao_cx_fixpoint_300x2000_d1.npy")).reshape(P, 3)
    obj, viol = ko.verify(t, X)
    print("incumbent", obj, "violations", int(np.asarray(viol).sum()))
    lp = klp.build(t)
    val, y, x, sec = klp.solve_highs(lp)
    a, l, g = klp.duals_to_alg(t, lp, y)
    print(f"LP value {val:.4f} ({sec:.1f} s), exact dual value at the rounded multipliers {klp.exact_dual_value(t, a, l, g):.4f}")
    reg, cfg = analyse(t, X, a, l, g)
    print("sum of regrets %.4f, partitions with regret > 1e-3: %d" % (reg.sum(), int((reg > 1e-3).sum())))
    rack = np.asarray(t.rack_of)
    nrep = np.bincount(X.reshape(-1), minlength=B); nlead = np.bincount(X[:, 0], minlength=B); nrack = np.bincount(rack[X.reshape(-1)], minlength=R)
    rep_lo, rep_hi = ko._floor_ceil(P * t.rf, B); lead_lo, lead_hi = ko._floor_ceil(P, B)
    af, lf, gf = a / 65536.0, l / 65536.0, g / 65536.0
    cs = 0.0
    for b in range(B):
        cs += af[b] * ((rep_hi if af[b] > 0 else rep_lo) - nrep[b]) + lf[b] * ((lead_hi if lf[b] > 0 else lead_lo) - nlead[b])
    print("complementary-slackness loss of the broker rows %.4f" % cs)
    print("bands: replicas", rep_lo, rep_hi, "leaders", lead_lo, lead_hi)
    for p in np.argsort(-reg)[:12]:
        if reg[p] <= 1e-3: break
        print(f"  partition {p}: regret {reg[p]:.3f} current {[int(b) for b in t.current[p]]} incumbent {[int(b) for b in X[p]]} priced best {list(cfg[p])}"
              f"  a {[round(float(af[b]),2) for b in X[p]]} l(lead) {round(float(lf[X[p][0]]),2)}")
    loose = [(b, int(nrep[b]), round(float(af[b]), 3)) for b in range(B) if abs(af[b]) > 1e-3 and nrep[b] != (rep_hi if af[b] > 0 else rep_lo)]
    loosel = [(b, int(nlead[b]), round(float(lf[b]), 3)) for b in range(B) if abs(lf[b]) > 1e-3 and nlead[b] != (lead_hi if lf[b] > 0 else lead_lo)]
    print("priced replica rows not at their end:", loose[:20])
    print("priced leader rows not at their end:", loosel[:20])
