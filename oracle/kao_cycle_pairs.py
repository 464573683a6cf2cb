"""KAO-CX extension, PROTOTYPE (analysis / next-round specification) -- TEST INFRASTRUCTURE ONLY, never imported by the product.

What KAO-CX (oracle/kao_cycle.py, kao_cycle.hip) still misses on rigid topics has one shape (local branching with HiGHS around
its fixpoints, tools/analysis/local_branch.py; drifted 300 x 2000, 14825 -> 14826: five partitions): a cycle of replica units
that runs THROUGH the leader slots of leader transfers whose leader units balance only pairwise.  Every piece is a primitive the
L layer has (role swap / promote / plain / demote / double), but the L layer prices each transfer with its own compensating F
path and uses a partition once.  Here leader-balanced PAIRS of transfers -- partition p led by u hands the leadership to v,
partition q led by v hands it to u -- are enumerated (every half-move of gain >= GMIN; a generic entering follower may take the
broker the partner releases), and a pair whose net replica effect is one unit x -> z becomes a COMPOUND EDGE of the F graph
(cost = -(gain of both rows), payload = the two new rows).  The closure of the augmented graph (same min-plus squarings) then has
negative diagonal entries where plain KAO-CX had none; cycles are unrolled (compound edges apply their two rows, F edges their
slot), evaluated exactly by the independent verifier, and handed back to KAO-CX.

Measured (scalar replay of K-search + oracle KAO-CX to a fixpoint, drifted 300 x 2000, MILP optimum 14826; six fixpoints below
it): 14825 -> 14826 twice, 14824 -> 14825, 14823 -> 14824, two unchanged (14823, 14825); every gain is "a cycle through 2 compound
edges"; seven more fixpoints (250 x 2000, optimum 14918: 14917, 14909, 14910; second drift seed of 300 x 2000, optimum 14801: 14797, 14796,
14798, 14799): 14910 -> 14914 in two passes, 14797 -> 14798, five unchanged.  ~70 s per pass in pure Python (324,000 half-moves, 580,000 pairs, 51,000 compound edges at B = 300): the enumeration is a
join on (leader, new leader) -- one wavefront per partition pair on the device.

usage: python oracle/kao_cycle_pairs.py B R P drift_seed fixpoint.npy     (the instance of synthetic.drift(make_cluster(B, R, 1, P, 3), 0.2, seed))
"""
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kao_cycle as kc   # noqa: E402
import kao_oracle as ko  # noqa: E402

ANY = -1
GMIN = -2

def half_moves(rd, p):
    """[(v, new_row, outs, ins, gain)] of partition p: every way to hand the leadership to another broker v, gain >= GMIN.
    ins may contain ANY (a generic entering follower: weight 0)."""
    t = rd.t; B = rd.B; RF = rd.RF
    row = [int(x) for x in rd.A[p]]
    u = row[0]
    WL, WF = rd.WL[p], rd.WF[p]
    w0 = rd.row_weight(p, row)
    ycur = [int(b) for b in rd.cur[p] if b < B and b not in row]
    out = []
    def add(v, new, outs, ins):
        g = rd.row_weight(p, [x if x != ANY else row[0] for x in new]) if ANY not in new else None
        if g is None:   # ANY enters as a follower with weight 0
            g = int(WL[new[0]]) + sum(int(WF[x]) for x in new[1:] if x != ANY)
        g -= w0
        if g >= GMIN: out.append((v, tuple(new), tuple(outs), tuple(ins), g))
    for k in range(1, RF):
        v = row[k]; others = [row[j] for j in range(1, RF) if j != k]
        add(v, [v, u] + others, (), ())                                   # a: role swap
        for y in ycur + [ANY]:
            add(v, [v, y] + others, (u,), (y,))                           # b: promote, u leaves, y enters
    # v outside the row: generic v only pays when the current leader is cheap to give up
    vs = set(b for b in (int(x) for x in rd.cur[p]) if b < B and b not in row)
    # the best any half-move towards a generic v (weight 0 as leader) can gain: skip the sweep over all brokers when even that is < GMIN
    wf_row = [int(WF[x]) for x in row]
    best_generic = max([sum(wf_row[1:])] + [int(WF[u]) + sum(wf_row[1:]) - wf_row[k] for k in range(1, RF)] +
                       [max([0] + [int(WF[y]) for y in ycur]) + sum(wf_row[1:]) - wf_row[k] for k in range(1, RF)]) - w0
    cand_v = range(B) if best_generic >= GMIN else sorted(vs)
    for v in cand_v:
        if v in row: continue
        add(v, [v] + row[1:], (u,), (v,))                                 # c: plain
        for k in range(1, RF):
            b = row[k]; others = [row[j] for j in range(1, RF) if j != k]
            add(v, [v, u] + others, (b,), (v,))                           # d: demote u, b leaves
            for y in ycur + [ANY]:
                if y == v: continue
                add(v, [v, y] + others, (u, b), (v, y))                   # e: double
    return u, out

def c7_ok(rd, row):
    if len(set(row)) != len(row): return False
    cnt = np.bincount(rd.rack[list(row)], minlength=rd.R)
    return not ((cnt > rd.bd["prack_hi"]).any() or (cnt < rd.bd["prack_lo"]).any())

def compound_edges(rd, verbose=True):
    """{(x, z): (cost, payload)}: x loses a replica unit, z gains one, leader counts unchanged; payload = [(p, row), (q, row)]."""
    t0 = time.time()
    buckets = defaultdict(list)
    n_h = 0
    for p in range(rd.P):
        u, hm = half_moves(rd, p)
        for (v, new, outs, ins, g) in hm:
            buckets[(u, v)].append((p, new, outs, ins, g)); n_h += 1
    edges = {}; closed = []
    n_pairs = 0
    for (u, v), hp in buckets.items():
        if u > v or (v, u) not in buckets: continue
        hq = buckets[(v, u)]
        for (p, newp, op, ip, gp) in hp:
            for (q, newq, oq, iq, gq) in hq:
                if p == q: continue
                n_pairs += 1
                outs = list(op) + list(oq); ins = list(ip) + list(iq)
                # cancel concrete brokers
                for x in list(outs):
                    if x in ins: outs.remove(x); ins.remove(x)
                # an ANY may take a leaving broker of the PARTNER (it enters the other partition)
                newp2, newq2 = list(newp), list(newq)
                def bind(newrow, own_outs, partner_outs):
                    nonlocal outs, ins
                    if ANY in newrow and ANY in ins:
                        for x in partner_outs:
                            if x in outs and x not in newrow:
                                cand = [x if w == ANY else w for w in newrow]
                                if c7_ok(rd, cand):
                                    outs.remove(x); ins.remove(ANY)
                                    return cand
                    return newrow
                newp2 = bind(newp2, op, oq); newq2 = bind(newq2, oq, op)
                if len(outs) > 1 or len(ins) > 1: continue
                gain = gp + gq
                if ANY in newp2 or ANY in newq2:
                    # one free entering broker left: an edge to every z that fits (z must keep C7 in its partition)
                    if len(outs) != 1 or ins != [ANY]: continue
                    x = outs[0]
                    who, rowt = (p, newp2) if ANY in newp2 else (q, newq2)
                    if ANY in newp2 and ANY in newq2: continue
                    for z in range(rd.B):
                        cand = [z if w == ANY else w for w in rowt]
                        if z == x or not c7_ok(rd, cand): continue
                        other = (q, newq2) if who == p else (p, newp2)
                        if not c7_ok(rd, other[1]): continue
                        key = (x, z); cost = -gain
                        if key not in edges or cost < edges[key][0]: edges[key] = (cost, [(who, cand), other])
                    continue
                if not (c7_ok(rd, newp2) and c7_ok(rd, newq2)): continue
                if not outs and not ins:
                    if gain > 0: closed.append((gain, [(p, newp2), (q, newq2)]))
                    continue
                if len(outs) == 1 and len(ins) == 1:
                    key = (outs[0], ins[0]); cost = -gain
                    if key not in edges or cost < edges[key][0]: edges[key] = (cost, [(p, newp2), (q, newq2)])
    if verbose: print(f"half-moves {n_h}, buckets {len(buckets)}, pairs {n_pairs}, compound edges {len(edges)}, closed improving pairs {len(closed)} ({time.time()-t0:.1f}s)", flush=True)
    return edges, closed

def find_improvement(t, X, verbose=True):
    rd = kc.Round(t, X)
    base, v0 = kc.evaluate(t, rd.A); assert v0 == 0
    edges, closed = compound_edges(rd, verbose)
    best = None
    def try_apply(parts, extra_path=None):
        Y = rd.A.copy(); used = set()
        for (p, row) in parts:
            if p in used: return None
            used.add(p); Y[p] = row
        if extra_path is not None:
            if not rd._walk(Y, used, 0, extra_path): return None
        o, v = kc.evaluate(t, Y)
        return (o, Y) if v == 0 and o > base else None
    for gain, parts in sorted(closed, key=lambda c: -c[0])[:50]:
        r = try_apply(parts)
        if r: return r[0], r[1], "closed pair"
    # augmented F graph: D = min(F edges, compound edges); closure; negative cycles
    n = rd.n
    D0 = rd.DF[0].copy()
    comp = np.zeros((n, n), dtype=bool)
    for (x, z), (cost, payload) in edges.items():
        if cost < D0[x, z]: D0[x, z] = cost; comp[x, z] = True
    Ds = [D0]; Ms = [None]
    for _ in range(kc.LEVELS):
        d, m = kc.Round._square(Ds[-1]); Ds.append(d); Ms.append(m)
    for lev in range(1, kc.LEVELS + 1):
        dg = np.diag(Ds[lev])[:rd.B]
        if (dg < 0).any():
            order = np.argsort(dg)
            for b in order[:40]:
                if dg[b] >= 0: break
                # unroll
                def path(u, v, lv):
                    if u == v: return [u]
                    if lv == 0: return [u, v]
                    m = int(Ms[lv][u, v]); a = path(u, m, lv - 1); c = path(m, v, lv - 1); return a + c[1:]
                m = int(Ms[lev][b, b])
                pth = path(b, m, lev - 1) + path(m, b, lev - 1)[1:]
                Y = rd.A.copy(); used = set(); ok = True
                for s, d in zip(pth[:-1], pth[1:]):
                    if s == d or s >= rd.B or d >= rd.B: continue
                    if comp[s, d]:
                        for (p, row) in edges[(s, d)][1]:
                            if p in used: ok = False; break
                            used.add(p); Y[p] = row
                        if not ok: break
                    else:
                        if not rd._walk(Y, used, 0, [s, d]): ok = False; break
                if not ok: continue
                o, v = kc.evaluate(t, Y)
                if v == 0 and o > base:
                    return o, Y, f"cycle level {lev} through {sum(bool(comp[s, d]) for s, d in zip(pth[:-1], pth[1:]))} compound edges, priced {-int(dg[b])}"
            if verbose: print(f"  level {lev}: {int((dg < 0).sum())} negative diagonal entries, none realised feasibly", flush=True)
    return None



def improve_with_pairs(t, X, max_passes=8, verbose=True):
    """KAO-CX fixpoint -> compound-edge pass -> KAO-CX ... until a pass finds nothing.  Returns (assignment, [objective per pass])."""
    cur, _ = kc.improve(t, X, max_rounds=100)
    objs = [kc.evaluate(t, cur)[0]]
    for _ in range(max_passes):
        r = find_improvement(t, cur, verbose)
        if r is None:
            break
        if verbose:
            print(f"  {objs[-1]} -> {r[0]} by {r[2]}", flush=True)
        cur, _ = kc.improve(t, r[1], max_rounds=100)
        objs.append(kc.evaluate(t, cur)[0])
    return cur, objs


def drift_topic(B, R, P, dseed):
    from kafka_assignment_optimizer_amd import synthetic as sy
    pt = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, dseed)[0]
    return ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                    n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current))


if __name__ == "__main__":
    B, R, P, d = (int(v) for v in sys.argv[1:5])
    t = drift_topic(B, R, P, d)
    X = np.load(sys.argv[5]).astype(np.uint16).reshape(P, 3)
    t0 = time.time()
    Y, objs = improve_with_pairs(t, X)
    print(f"B={B} P={P} d{d}: objectives per pass {objs} ({time.time() - t0:.0f}s)")
