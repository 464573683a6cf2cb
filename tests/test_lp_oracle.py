"""KAO-LP oracle (round 5; CPU): the compact LP relaxation (oracle/kao_lp.py), its HiGHS reference, the generic-sparse
restatement of the interior-point iteration and the block-structured C restatement (oracle/kao_lp_port.c) agree, and the exact
Lagrangian dual value at the LP's row duals (oracle/kao_port.c) is the LP value -- the certificate the device reports."""
import math

import numpy as np
import pytest

from conftest import load_golden


def _drift_topic(ko, B, R, P, dseed=1):
    from kafka_assignment_optimizer_amd import synthetic as sy
    pt = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, dseed)[0]
    return ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                    n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights,
                    bounds_override=dict(pt.bounds_override))


def test_kat1_compact_lp(ko, kp):
    """README.md:52-63 -> README.md:85-91: the compact LP of the worked example has the value 58 (= the optimum) under HiGHS,
    under the generic iteration and under the structured one, and the exact dual value at its duals is 58."""
    import kao_lp as kl
    t = ko.readme_example()
    lp = kl.build(t)
    val, y, _, _ = kl.solve_highs(lp)
    assert abs(val - 58.0) < 1e-6
    a, l, g = kl.duals_to_alg(t, lp, y)
    assert math.floor(kl.exact_dual_value(t, a, l, g) + 1e-9) == 58
    r = kl.port_solve(t)
    assert r["status"] == 0 and abs(r["dual"] - 58.0) < 1e-4 and abs(r["primal"] - 58.0) < 1e-4
    assert math.floor(kl.exact_dual_value(t, r["a"], r["l"], r["g"]) + 1e-9) == 58


@pytest.mark.parametrize("B,R,P", [(100, 5, 1000), (130, 5, 1000)])
def test_structured_iteration_matches_the_generic_one(ko, kp, B, R, P):
    """Rigid bands (100 x 1000: two exactly dependent coupling rows are pinned) and slack bands (130 x 1000): the block
    elimination of oracle/kao_lp_port.c follows the generic-sparse iteration of oracle/kao_lp.py iterate by iterate
    (mu, primal and dual objective to 1e-6 relative while mu >= 1e-6), both end at the HiGHS value, and the exact dual value
    at the structured iteration's duals is floor-equal to it (tests/golden/drift_scale.json: 100 x 1000 -> 7430)."""
    import kao_lp as kl
    t = _drift_topic(ko, B, R, P)
    lp = kl.build(t)
    val, _, _, _ = kl.solve_highs(lp)
    tr = []
    _, y, it, po, do = kl.ipm(lp, trace=tr)
    r = kl.port_solve(t)
    assert r["status"] == 0 and abs(r["iterations"] - it) <= 1
    for (m0, p0, d0, _, _), (m1, p1, d1, _, _) in zip(tr, r["trace"]):
        if m0 < 1e-6:
            break
        assert abs(m0 - m1) <= 1e-6 * m0 and abs(p0 - p1) <= 1e-6 * abs(p0) and abs(d0 - d1) <= 1e-6 * abs(d0)
    assert abs(do - val) < 1e-3 and abs(r["dual"] - val) < 1e-3
    L = kl.exact_dual_value(t, r["a"], r["l"], r["g"])
    assert math.floor(L + 1e-9) == round(val)
    if (B, P) == (100, 1000):
        gold = [x for x in load_golden("drift_scale.json")["rows"] if (x["B"], x["P"]) == (100, 1000)][0]
        assert round(val) == gold["milp_objective"] == 7430


def test_structured_iteration_on_the_golden_families(ko, kp):
    """RF 5..8 (two replicas per rack: the C5 rows and bounded C7 slacks are live) and the medium family (one rack, RF = R: the
    C7 rows add up to C1 and T is singular -- guarded pivots): the structured iteration converges on every feasible instance
    and floor(exact dual value at its duals) is never below the HiGHS optimum and equals it on all but a few (LP gaps)."""
    import kao_lp as kl
    n = exact = 0
    for c in load_golden("random_rf.json")["cases"][:60]:
        if c["status"] != "optimal":
            continue
        t = ko.random_case_rf(c["seed"])
        r = kl.port_solve(t)
        assert r["status"] == 0, c["seed"]
        b = math.floor(kl.exact_dual_value(t, r["a"], r["l"], r["g"]) + 1e-9)
        assert b >= c["objective"], c["seed"]
        n += 1; exact += b == c["objective"]
    for c in load_golden("random_medium.json")["cases"][:50]:
        if c["status"] != "optimal":
            continue
        t = ko.topic_from_dict(c["topic"])
        r = kl.port_solve(t)
        assert r["status"] == 0, c["seed"]
        b = math.floor(kl.exact_dual_value(t, r["a"], r["l"], r["g"]) + 1e-9)
        assert b >= c["objective"], c["seed"]
        n += 1; exact += b == c["objective"]
    assert n >= 60 and exact >= n - 2, (n, exact)


@pytest.mark.parametrize("B,R,P,dseed,optimum", [(100, 5, 1000, 1, 7430), (300, 6, 2000, 1, 14826), (300, 6, 2000, 2, 14801)])
def test_perturbed_lp_rounds_to_the_milp_optimum(ko, kp, B, R, P, dseed, optimum):
    """KAO-LP's primal side (round 5).  The optimal face of the model's LP (README.md:144-185 relaxed) is huge and the
    interior-point iterate ends at its centre; with costs perturbed by eps * hash(variable) the iterate converges to ONE vertex,
    and that vertex rounds (oracle/kao_lp.py round_primal) to an assignment the README's rows accept (ko.verify) whose objective
    is the HiGHS MILP optimum of tests/golden/drift_scale.json (7430, 14826) / the second drift seed's (14801, the one K-search +
    KAO-CX miss on one solver seed in five) -- for two salts.  A SMALL perturbation (what kao_solve uses) also keeps the
    certificate: the exact dual value at its row duals still floors to the optimum."""
    import kao_lp as kl
    t = _drift_topic(ko, B, R, P, dseed)
    for salt in (0, 1):
        r = kl.port_solve(t, tol=1e-8, maxit=150, primal=True, pert=kl.default_pert(t), salt=salt)
        assert r["status"] == 0
        A, rep = kl.round_primal(t, *kl.primal_blocks(t, r["x"], r["xg"]))
        obj, viol = ko.verify(t, A)
        assert int(np.asarray(viol).sum()) == 0 and obj == optimum, (salt, obj, rep)
        assert rep["over_inflow"] == 0 and rep["unplaced"] == 0
    eps = min(1e-4, 1.5 / (P * 3))
    r = kl.port_solve(t, tol=1e-10, maxit=200, primal=True, pert=eps)
    assert r["status"] == 0 and math.floor(kl.exact_dual_value(t, r["a"], r["l"], r["g"]) + 1e-9) == optimum
    A, rep = kl.round_primal(t, *kl.primal_blocks(t, r["x"], r["xg"]))
    obj, viol = ko.verify(t, A)
    assert int(np.asarray(viol).sum()) == 0 and obj == optimum, (obj, rep)


def test_round_primal_without_a_vertex(ko, kp):
    """An iterate far from a vertex (six iterations): most partitions are fractional.  Without a fallback they take their heaviest
    options -- every row complete, leader first, no broker twice, one replica per rack where the model says so (rows C1, C2, C5,
    C7 of README.md:148-180 hold; the band rows need not) --, with one they keep the fallback's rows."""
    import kao_lp as kl
    t = _drift_topic(ko, 100, 5, 1000)
    r = kl.port_solve(t, tol=1e-8, maxit=6, primal=True, pert=kl.default_pert(t))
    blocks = kl.primal_blocks(t, r["x"], r["xg"])
    A, rep = kl.round_primal(t, *blocks)
    assert rep["fractional"] > 100
    _, viol = ko.verify(t, A)
    v = [int(x) for x in np.asarray(viol)]
    assert v[1] == 0 and v[2] == 0 and v[5] == 0 and v[7] == 0, v     # C1, C2, C5, C7
    fb = np.tile(np.arange(3), (1000, 1))
    A2, rep2 = kl.round_primal(t, *blocks, fallback=fb)
    assert rep2["from_fallback"] == rep2["fractional"] == rep["fractional"]
    assert int((A2 == fb).all(axis=1).sum()) >= rep2["fractional"]


def _pack(F, L, YF, YL, ZF, ZL):
    return np.concatenate([F.T, L.T, YF.T, YL.T], axis=0).astype(np.uint8), np.concatenate([ZF, ZL]).astype(np.int32)


def test_host_rounding_matches_the_specification(ko, kp):
    """The product's host half of kao_lp_round (kao_round.cpp, reached through the test hook kao_lp_round_host: no device involved)
    against the specification (oracle/kao_lp.py round_primal) on the scalar restatement's iterate: the same assignment, the same
    counts -- RF 5..8 (two replicas per rack: C5 rows), the medium family, a drifted 1,000-partition topic; at a vertex (swaps, the
    bounded search over the fractional partitions) and after five iterations (hundreds of fractional partitions: the greedy pass;
    with and without fallback rows)."""
    import kao_lp as kl
    import kafka_assignment_optimizer_amd as kao
    from conftest import to_product_topic
    cases = [ko.random_case_rf(c["seed"]) for c in load_golden("random_rf.json")["cases"] if c["status"] == "optimal"][::2]
    cases += [ko.topic_from_dict(c["topic"]) for c in load_golden("random_medium.json")["cases"] if c["status"] == "optimal"][::2]
    cases.append(_drift_topic(ko, 100, 5, 1000))
    n = n_search = n_swaps = 0
    for t in cases:
        for maxit in (150, 5):
            r = kl.port_solve(t, tol=1e-8, maxit=maxit, primal=True, pert=kl.default_pert(t))
            blocks = kl.primal_blocks(t, r["x"], r["xg"])
            A, rep = kl.round_primal(t, *blocks)
            q, zq = _pack(*blocks)
            d = kao.lp_round_host(to_product_topic(t), q, zq)
            assert d["assignment"].tolist() == A.tolist(), (t.name, maxit, rep)
            assert (d["fractional"], d["over_inflow"], d["unplaced"]) == (rep["fractional"], rep["over_inflow"], rep["unplaced"])
            n += 1; n_search += 0 < rep["fractional"] <= kl.MAX_SEARCH; n_swaps += rep["swaps"] > 0
    t = cases[-1]
    r = kl.port_solve(t, tol=1e-8, maxit=5, primal=True, pert=kl.default_pert(t))
    blocks = kl.primal_blocks(t, r["x"], r["xg"])
    fb = np.tile(np.array([97, 98, 99]), (t.n_partitions, 1))
    A, rep = kl.round_primal(t, *blocks, fallback=fb)
    d = kao.lp_round_host(to_product_topic(t), *_pack(*blocks), fallback=fb)
    assert d["assignment"].tolist() == A.tolist() and d["from_fallback"] == rep["from_fallback"] == rep["fractional"] > 100
    assert n >= 130 and n_search >= 5 and n_swaps >= 3, (n, n_search, n_swaps)


@pytest.mark.parametrize("B,R,P,dseed,salt,fractional", [(60, 6, 400, 1, 1, 14), (100, 10, 1000, 2, 2, 9), (300, 10, 2000, 1, 3, 13),
                                                         (300, 10, 2000, 2, 4, 19)])
def test_half_integral_vertex_is_completed_by_patterns(ko, kp, monkeypatch, B, R, P, dseed, salt, fractional):
    """Perturbed LPs whose vertex is half-integral in 9..19 partitions (rigid bands: every broker's band is a single value).  The
    pattern completion (oracle/kao_lp.py complete_by_patterns: which current replicas each of these partitions keeps, heaviest
    first, bounded by the best remaining patterns and by the room the brokers' bands have left -- the 19-partition case needs the
    second bound: 126 nodes instead of 218,000; the new replicas matched to the brokers still below their band) gives an assignment the README's rows accept
    (README.md:148-180, ko.verify) whose weight IS the certificate, floor of the exact dual value at the unperturbed LP's row
    duals -- optimal, proven.  The search over candidate rows it replaced ended 2..12 below on the same iterates (docs/notes_r05.md
    section 6).  The product's host half (kao_round.cpp) returns the same rows."""
    import kao_lp as kl
    import kafka_assignment_optimizer_amd as kao
    from conftest import to_product_topic
    # the iterates are this test's INPUT (picked because they end half-integral): they were produced with the step fraction 0.9995; with
    # today's 0.9 the same solves stop on iterates that round without a fractional partition.  The hook keeps the pinned inputs.
    monkeypatch.setenv("KAO_LP_GAMMA", "0.9995"); monkeypatch.setenv("KAO_LP_SIGEXP", "3"); monkeypatch.setenv("KAO_LP_XFLOOR", "1")      # (likewise the centering exponent, 3 then, 10 today, and the starting point's floor, 1 then, 0.1 today)
    t = _drift_topic(ko, B, R, P, dseed)
    r0 = kl.port_solve(t)
    bound = math.floor(kl.exact_dual_value(t, r0["a"], r0["l"], r0["g"]) + 1e-9)
    r = kl.port_solve(t, tol=1e-8, maxit=200, primal=True, pert=min(1e-2, 100.0 / (P * 3)), salt=salt)
    assert r["status"] == 0
    blocks = kl.primal_blocks(t, r["x"], r["xg"])
    A, rep = kl.round_primal(t, *blocks)
    obj, viol = ko.verify(t, A)
    assert rep["fractional"] == fractional and rep["patterns"] == 1, rep
    assert int(np.asarray(viol).sum()) == 0 and obj == bound, (obj, bound, rep)
    d = kao.lp_round_host(to_product_topic(t), *_pack(*blocks))
    assert d["assignment"].tolist() == A.tolist() and d["fractional"] == fractional


def test_rows_outside_the_inflows_or_over_a_band_join_the_pattern_completion(ko, kp, monkeypatch):
    """400 x 6000, second drift seed, tolerance 1e-6: three fractional partitions and ONE row of the integral pass that found no
    broker with inflow left in its rack (`over_inflow` 1: broker 0 ends a replica over its band, ten brokers one under it with nine
    slots to give -- no completion can be perfect, and the band repair that followed cost nine units).  That row is given up and
    completed with the fractional ones: the certificate (45366, floor of the exact dual value at the unperturbed LP's row duals),
    no violation, no repair; the product's host half returns the same rows and counts."""
    import kao_lp as kl
    import kafka_assignment_optimizer_amd as kao
    from conftest import to_product_topic
    monkeypatch.setenv("KAO_LP_GAMMA", "0.9995"); monkeypatch.setenv("KAO_LP_SIGEXP", "3"); monkeypatch.setenv("KAO_LP_XFLOOR", "1")     # pinned input iterates (see test_half_integral_vertex_is_completed_by_patterns)
    t = _drift_topic(ko, 400, 8, 6000, 2)
    r0 = kl.port_solve(t)
    bound = math.floor(kl.exact_dual_value(t, r0["a"], r0["l"], r0["g"]) + 1e-9)
    r = kl.port_solve(t, tol=1e-6, maxit=200, primal=True, pert=min(1e-2, 100.0 / (6000 * 3)), salt=2)
    blocks = kl.primal_blocks(t, r["x"], r["xg"])
    A, rep = kl.round_primal(t, *blocks)
    obj, viol = ko.verify(t, A)
    assert (rep["fractional"], rep["over_inflow"], rep["patterns"], rep["repaired"]) == (4, 1, 1, 0), rep
    assert int(np.asarray(viol).sum()) == 0 and obj == bound == 45366, (obj, bound)
    d = kao.lp_round_host(to_product_topic(t), *_pack(*blocks))
    assert d["assignment"].tolist() == A.tolist() and (d["fractional"], d["over_inflow"]) == (4, 1)
    # the same topic stopped at tolerance 1e-5: 46 fractional partitions, and the rows already set hold one broker 46 times (band:
    # 45) although every replica stayed inside the inflows -- 139 places to fill with 138 slots.  The last row holding that broker
    # joins the completion (47 partitions): two under the certificate without a violation or a repair (candidate rows + repair: 50).
    r = kl.port_solve(t, tol=1e-5, maxit=200, primal=True, pert=min(1e-2, 100.0 / (6000 * 3)), salt=2)
    blocks = kl.primal_blocks(t, r["x"], r["xg"])
    A, rep = kl.round_primal(t, *blocks)
    obj, viol = ko.verify(t, A)
    assert (rep["fractional"], rep["over_inflow"], rep["patterns"], rep["repaired"]) == (47, 0, 1, 0), rep
    assert int(np.asarray(viol).sum()) == 0 and bound - 2 <= obj <= bound, (obj, bound)
    d = kao.lp_round_host(to_product_topic(t), *_pack(*blocks))
    assert d["assignment"].tolist() == A.tolist() and d["fractional"] == 47


def test_band_repair_matches_the_specification(ko, kp):
    """The band repair at the end of the rounding (kao_round.cpp against oracle/kao_lp.py repair_bands, through kao_lp_round_host mode 2),
    on imbalances built from an optimal assignment of a drifted 100-broker topic: (a) weightless follower replicas piled onto two
    brokers (zero-cost moves, same rack first), (b) kept current replicas given up elsewhere and re-added on another broker (only
    costly moves are left), (c) leaderships shifted between brokers of the same partitions (role swaps) and (d) between brokers that
    share no partition (a chain of role swaps).  Same assignment from both, and the README's band rows hold again (ko.verify)."""
    import kao_lp as kl
    import kafka_assignment_optimizer_amd as kao
    from conftest import to_product_topic
    t = _drift_topic(ko, 100, 5, 1000)
    r = kl.port_solve(t, tol=1e-8, maxit=150, primal=True, pert=kl.default_pert(t))
    A0, _ = kl.round_primal(t, *kl.primal_blocks(t, r["x"], r["xg"]))
    obj0, v0 = ko.verify(t, A0)
    assert int(np.asarray(v0)[0]) == 0
    pt = to_product_topic(t)
    rack = np.asarray(t.rack_of)
    cur = np.asarray(t.current)
    rng = np.random.default_rng(11)

    def check(A, what, expect_feasible=True):
        ref = A.copy()
        moves = kl.repair_bands(t, ref)
        got = kao.lp_repair_host(pt, A)
        assert got.tolist() == ref.tolist(), what
        o, v = ko.verify(t, ref)
        if expect_feasible:
            assert int(np.asarray(v)[0]) == 0 and moves > 0, (what, [int(x) for x in np.asarray(v)], moves)
        return o

    # (a) weightless followers piled onto a broker of the same rack
    A = A0.copy(); n = 0
    for p in range(t.n_partitions):
        for k in (1, 2):
            b = int(A[p, k])
            if b in cur[p]: continue
            tgt = next((x for x in range(t.n_brokers) if rack[x] == rack[b] and x != b and x not in A[p] and x not in cur[p]), None)
            if tgt is not None and n < 3: A[p, k] = tgt; n += 1
        if n >= 3: break
    assert check(A, "weightless followers") == obj0
    # (b) kept current followers moved to brokers that are not current replicas: the repair has to pay for moving others, or move them on
    A = A0.copy(); n = 0
    for p in range(t.n_partitions):
        for k in (1, 2):
            b = int(A[p, k])
            if b not in cur[p]: continue
            tgt = next((x for x in range(t.n_brokers) if rack[x] == rack[b] and x != b and x not in A[p] and x not in cur[p]), None)
            if tgt is not None and n < 2: A[p, k] = tgt; n += 1
        if n >= 2: break
    assert check(A, "costly followers") <= obj0
    # (c) role swaps inside partitions whose two brokers carry no weight there
    A = A0.copy(); n = 0
    for p in range(t.n_partitions):
        if n >= 2: break
        for k in (1, 2):
            if int(A[p, 0]) not in cur[p] and int(A[p, k]) not in cur[p]:
                A[p, 0], A[p, k] = A[p, k], A[p, 0]; n += 1; break
    check(A, "role swaps")
    # (d) a leadership moved between two brokers that share no partition (300 brokers: most pairs do not): only a chain of role swaps
    #     brings it back -- u's leadership of p goes to v, v's leadership of another partition q goes on to w
    t2 = _drift_topic(ko, 300, 6, 2000)
    r2 = kl.port_solve(t2, tol=1e-8, maxit=150, primal=True, pert=kl.default_pert(t2))
    B0, _ = kl.round_primal(t2, *kl.primal_blocks(t2, r2["x"], r2["xg"]))
    assert int(np.asarray(ko.verify(t2, B0)[1])[0]) == 0
    pt2 = to_product_topic(t2)
    lead_of = {}
    for p in range(t2.n_partitions): lead_of.setdefault(int(B0[p, 0]), []).append(p)
    holds = [set() for _ in range(t2.n_brokers)]
    for p in range(t2.n_partitions):
        for b in B0[p]: holds[int(b)].add(p)
    nlead = np.bincount(B0[:, 0], minlength=t2.n_brokers)
    llo2, lhi2 = t2.bounds()["lead_lo"], t2.bounds()["lead_hi"]      # u drops below the band, w rises above it
    A = None
    for p in range(t2.n_partitions):
        for k in (1, 2):
            u, v = int(B0[p, 0]), int(B0[p, k])
            for q in lead_of.get(v, []):
                for k2 in (1, 2):
                    w = int(B0[q, k2])
                    if q == p or w == u or holds[u] & holds[w] or nlead[u] != llo2 or nlead[w] != lhi2: continue
                    A = B0.copy()
                    A[p, 0], A[p, k] = v, u
                    A[q, 0], A[q, k2] = w, v
                    break
                if A is not None: break
            if A is not None: break
        if A is not None: break
    assert A is not None
    ref = A.copy()
    assert kl.repair_bands(t2, ref) >= 2                     # at least two swaps: a chain
    assert kao.lp_repair_host(pt2, A).tolist() == ref.tolist()
    assert int(np.asarray(ko.verify(t2, ref)[1])[0]) == 0
    # (e) round 6: a rigid rack band under loose broker bands (130 brokers, 5 racks: 23..24 replicas a broker, 600 a rack exactly) --
    #     followers moved to another rack leave the broker bands alone and two racks one replica off: the rack repair (repair_racks) moves one back
    t3 = _drift_topic(ko, 130, 5, 1000)
    bd3 = t3.bounds()
    assert bd3["rep_lo"] < bd3["rep_hi"] and bd3["rack_lo"] == bd3["rack_hi"]
    r3 = kl.port_solve(t3, tol=1e-8, maxit=150, primal=True, pert=kl.default_pert(t3))
    C0, _ = kl.round_primal(t3, *kl.primal_blocks(t3, r3["x"], r3["xg"]))
    obj3, v3 = ko.verify(t3, C0)
    assert int(np.asarray(v3)[0]) == 0
    pt3 = to_product_topic(t3)
    rack3 = np.asarray(t3.rack_of); cur3 = np.asarray(t3.current)
    for n_moves in (1, 2):
        A = C0.copy(); n = 0
        load3 = np.bincount(A.reshape(-1), minlength=t3.n_brokers)
        for p in range(t3.n_partitions):
            if n >= n_moves: break
            for k in (1, 2):
                b = int(A[p, k])
                if load3[b] <= bd3["rep_lo"]: continue
                tgt = next((x for x in range(t3.n_brokers) if rack3[x] != rack3[b] and load3[x] < bd3["rep_hi"] and x not in A[p] and x not in cur3[p]
                            and sum(1 for y in A[p] if rack3[y] == rack3[x]) < bd3["prack_hi"]), None)
                if tgt is None: continue
                A[p, k] = tgt; load3[b] -= 1; load3[tgt] += 1; n += 1
                break
        assert n == n_moves
        vb = np.asarray(ko.verify(t3, A)[1])
        assert int(vb[0]) == int(vb[6]) > 0, [int(x) for x in vb]      # rack rows (C6) only
        ref = A.copy()
        assert kl.repair_bands(t3, ref) == kl.repair_racks(t3, A.copy()) >= 1
        assert kao.lp_repair_host(pt3, A).tolist() == ref.tolist()
        o3, v3 = ko.verify(t3, ref)
        assert int(np.asarray(v3)[0]) == 0 and o3 <= obj3


def test_simplex_vertex_of_the_compact_lp_rounds_to_the_milp_optimum(ko, kp):
    """The observation KAO-LP's primal side rests on, pinned against an independent solver: a VERTEX of the compact LP (HiGHS dual
    simplex, scipy) is integral on the drifted 100 x 1000 topic, and the specification's rounding turns it into an assignment that the
    README's rows accept and whose objective is the HiGHS MILP optimum of the full model (7430, tests/golden/drift_scale.json).  (The
    interior-point restatement reaches a vertex through the cost perturbation; this test does not use it.)"""
    import kao_lp as kl
    t = _drift_topic(ko, 100, 5, 1000)
    lp = kl.build(t)
    assert kl.compact_index(t)["n"] == len(lp.c)
    val, _, x, _ = kl.solve_highs(lp, method="highs-ds")
    assert abs(val - 7430.0) < 1e-6
    assert int((np.abs(x - np.rint(x)) > 1e-6).sum()) == 0
    A, rep = kl.round_primal(t, *kl.blocks_from_compact(t, x))
    obj, viol = ko.verify(t, A)
    assert (rep["fractional"], rep["over_inflow"], rep["unplaced"]) == (0, 0, 0)
    assert int(np.asarray(viol)[0]) == 0 and obj == 7430
