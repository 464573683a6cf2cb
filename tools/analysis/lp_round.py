"""CPU (round 5): from a solution of the compact LP (oracle/kao_lp.py) to an assignment.
The compact LP pools the NEW replicas of a partition per rack (yf[p, r], yl[p, r]) and counts what every broker receives (zf[b],
zl[b]); an integral solution therefore still has to hand the new replicas of a rack to that rack's brokers -- any way that respects
the inflows and puts no broker twice into a partition (row C5, README.md:168-171) is as good as any other, the objective only
sees current replicas.  `assignment_from_compact` does that greedily (partitions in order, brokers by remaining inflow) and reports
what it could not place; fractional partitions (any variable farther than `tol` from an integer) keep the rows of `fallback`."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kao_oracle as ko
import kao_lp as klp


def compact_index(t):
    """Variable indices of kao_lp.build(t), by replaying its loops: dict(zf[B], zl[B], f[P][rf_cur], l[P][rf_cur], yf[P][R], yl[P][R]) (-1 = absent)."""
    B, R, P = t.n_brokers, t.n_racks, t.n_partitions
    bd = t.bounds()
    lo, hi, llo, lhi = bd["rep_lo"], bd["rep_hi"], bd["lead_lo"], bd["lead_hi"]
    rlo, rhi, plo, phi = bd["rack_lo"], bd["rack_hi"], bd["prack_lo"], bd["prack_hi"]
    has_c5, has_n = phi >= 2, hi > lo
    n = 0
    zf = np.zeros(B, int); zl = np.zeros(B, int)
    for b in range(B):
        zf[b] = n; zl[b] = n + 1; n += 2
        if has_n: n += 1
        if lhi > llo: n += 1
    if has_n and rhi > rlo: n += R
    f = -np.ones((P, t.rf_cur), int); l = -np.ones((P, t.rf_cur), int); yf = np.zeros((P, R), int); yl = np.zeros((P, R), int)
    for p in range(P):
        for j in range(t.rf_cur):
            b = int(t.current[p, j])
            if b == ko.NONE or b >= B: continue
            f[p, j] = n; l[p, j] = n + 1; n += 2
            if has_c5: n += 1
        for r in range(R):
            yf[p, r] = n; yl[p, r] = n + 1; n += 2
            if phi > plo: n += 1
    return dict(zf=zf, zl=zl, f=f, l=l, yf=yf, yl=yl, n=n)


def assignment_from_compact(t, F, L, YF, YL, ZF, ZL, fallback=None, tol=1e-6):
    """F, L [P][rf_cur], YF, YL [P][R], ZF, ZL [B] (floats).  Returns (A [P][RF] or None where a row is incomplete, report)."""
    B, R, P, RF = t.n_brokers, t.n_racks, t.n_partitions, t.rf
    rack = np.asarray(t.rack_of)
    members = [np.nonzero(rack == r)[0] for r in range(R)]
    capf = np.rint(ZF).astype(int).copy(); capl = np.rint(ZL).astype(int).copy()
    A = -np.ones((P, RF), int)
    frac_p = []; unplaced = 0; over = 0
    def isint(v): return np.all(np.abs(v - np.rint(v)) <= tol)
    order_f = [sorted(m, key=lambda b: b) for m in members]
    ptr = [0] * R
    for p in range(P):
        if not (isint(F[p]) and isint(L[p]) and isint(YF[p]) and isint(YL[p])):
            frac_p.append(p)
            if fallback is not None: A[p] = fallback[p]
            continue
        row = []; lead = -1
        for j in range(t.rf_cur):
            b = int(t.current[p, j])
            if b == ko.NONE or b >= B: continue
            if L[p, j] > 0.5: lead = b
            elif F[p, j] > 0.5: row.append(b)
        used = set(row) | ({lead} if lead >= 0 else set())
        def take(r, cap):
            nonlocal unplaced, over
            m = members[r]
            cand = [b for b in m if cap[b] > 0 and b not in used]
            if cand:
                b = max(cand, key=lambda b: (cap[b], -b)); cap[b] -= 1; return int(b)
            cand = [b for b in m if b not in used]
            if not cand: unplaced += 1; return -1
            over += 1
            return int(cand[0])
        for r in range(R):
            if YL[p, r] > 0.5:
                b = take(r, capl)
                if b >= 0: lead = b; used.add(b)
            for _ in range(int(round(YF[p, r]))):
                b = take(r, capf)
                if b >= 0: row.append(b); used.add(b)
        if lead < 0 or len(row) != RF - 1:
            frac_p.append(p)
            if fallback is not None: A[p] = fallback[p]
            continue
        A[p, 0] = lead; A[p, 1:] = row
    return A, dict(fractional_partitions=len(frac_p), unplaced=unplaced, over_inflow=over, left_f=int(capf.sum()), left_l=int(capl.sum()), frac_list=frac_p)


if __name__ == "__main__":
    import time
    from scipy.optimize import linprog
    from regret import topic
    B, R, P = [int(x) for x in sys.argv[1:4]]
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    t = topic(B, R, P, seed=seed)
    lp = klp.build(t)
    ix = compact_index(t)
    assert ix["n"] == len(lp.c), (ix["n"], len(lp.c))
    bounds = [(0.0, None if not np.isfinite(ub) else ub) for ub in lp.u]
    t0 = time.time()
    res = linprog(lp.c, A_eq=lp.A, b_eq=lp.b, bounds=bounds, method="highs-ds")
    x = res.x
    print(f"{B}x{P} d{seed}: simplex vertex value {-res.fun:.4f} in {time.time() - t0:.1f} s, fractional variables {int((np.abs(x - np.rint(x)) > 1e-6).sum())} of {len(x)}", flush=True)
    g = lambda idx: np.where(idx >= 0, x[np.maximum(idx, 0)], 0.0)
    A, rep = assignment_from_compact(t, g(ix["f"]), g(ix["l"]), g(ix["yf"]), g(ix["yl"]), g(ix["zf"]), g(ix["zl"]))
    print({k: v for k, v in rep.items() if k != "frac_list"})
    if (A >= 0).all():
        obj, viol = ko.verify(t, A)
        print("assignment: objective", obj, "violations", [int(v) for v in np.asarray(viol)])
