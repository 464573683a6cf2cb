"""KAO-CX on the MI355X through the C ABI (kao_improve_cycles / kao_cycle_matrices / kao_cycle_seeds) against
oracle/kao_cycle.py: bit-exact transfer-graph edges, closures and midpoints, seed tables, whole rounds and fixpoints; the
improved assignments stay feasible under the independent verifier; kao_solve with KAO-CX on a drifted 1000-partition topic."""
import numpy as np
import pytest

from conftest import load_golden, to_product_topic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kao():
    import kafka_assignment_optimizer_amd as k
    k.init(0)
    return k


def _cases(ko, kp, seeds, launches, iters):
    for s in seeds:
        t = ko.random_case_wide(s)
        if t.rf < 2 or t.rf > 4:
            continue
        r = kp.port_search(t, 3, 0, launches, iters)
        if r["best_obj"] >= 0:
            yield s, t, r["best"]


def _drifted(ko, B, R, P, rf=3, seed=1):
    from kafka_assignment_optimizer_amd import synthetic as sy
    pt = sy.drift(sy.make_cluster(B, R, 1, P, rf, [], []), 0.2, seed)[0]
    ot = ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                  n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights)
    return pt, ot


def _same_matrices(kao, kc, pt, t, a):
    rd = kc.Round(t, a)
    B = t.n_brokers
    for layer, (Ds, Ms, E) in enumerate(((rd.DF, rd.MF, rd.EF), (rd.DS, rd.MS, rd.ES), (rd.DL, rd.ML, rd.EL))):
        for lev in range(4):
            d, m, s = kao.cycle_matrices(pt, a, layer, lev)
            assert np.array_equal(d, Ds[lev]), (layer, lev)
            if lev:
                assert np.array_equal(m, Ms[lev]), (layer, lev)
            else:
                es = (E & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                assert np.array_equal(s[:B, :B], es[:B, :B]), layer
    return rd


def test_matrices_seeds_and_rounds_match_the_oracle_on_the_wide_family(kao, ko, kp):
    import kao_cycle as kc
    n = n_seed_tables = n_improved = 0
    for launches, iters, seeds in ((6, 256, range(0, 50)), (1, 48, range(100, 160))):
        for s, t, a in _cases(ko, kp, seeds, launches, iters):
            pt = to_product_topic(t)
            rd = _same_matrices(kao, kc, pt, t, a)
            if not rd.cycle_candidates():
                assert np.array_equal(kao.cycle_seeds(pt, a), rd.seed_table()), s
                n_seed_tables += 1
            Xo, hist = kc.improve(t, a, 64)
            Xg, obj, st = kao.improve_cycles(pt, a, 64)
            assert np.array_equal(np.asarray(Xo).reshape(-1), Xg.reshape(-1)), s
            o, v = ko.verify(t, Xg)
            assert int(np.asarray(v).sum()) == 0 and o == obj == st["objective_after"]
            assert st["rounds"] == len(hist) and st["improving_rounds"] == sum("objective" in h for h in hist)
            n += 1
            n_improved += obj > st["objective_before"]
    assert n >= 25 and n_seed_tables >= 10 and n_improved >= 5


@pytest.mark.parametrize("shape", [(60, 3, 200, 3), (45, 5, 130, 2), (50, 2, 120, 3), (48, 6, 160, 4)])
def test_drifted_topics_match_the_oracle(kao, ko, kp, shape):
    """Rigid (60 x 200: every broker count fixed) and slack bands (45 brokers, RF 2), prack_lo = 1 (two racks, RF 3), RF 2 and 4."""
    import kao_cycle as kc
    B, R, P, rf = shape
    pt, t = _drifted(ko, B, R, P, rf)
    r = kp.port_search(t, 3, 0, 2, 64)
    assert r["best_obj"] >= 0
    a = r["best"]
    rd = _same_matrices(kao, kc, pt, t, a)
    if not rd.cycle_candidates():
        assert np.array_equal(kao.cycle_seeds(pt, a), rd.seed_table())
    Xo, hist = kc.improve(t, a, 64)
    Xg, obj, st = kao.improve_cycles(pt, a, 64)
    assert np.array_equal(np.asarray(Xo).reshape(-1), Xg.reshape(-1))
    o, v = ko.verify(t, Xg)
    assert int(np.asarray(v).sum()) == 0 and o == obj >= st["objective_before"]


@pytest.mark.parametrize("shape", [(60, 3, 200, 3), (45, 5, 130, 2), (48, 6, 160, 4), (100, 5, 1000, 3)])
def test_bulk_mode_matches_the_oracle(kao, ko, kp, shape, monkeypatch):
    """Round 4: the bulk mode of a KAO-CX round (topics beyond 131,072 replica slots: cycle candidates are merged before they
    are scored) forced onto small topics (test hook KAO_CX_BULK_SLOTS=0; oracle round_step(bulk_slots=0)): same assignment
    after every call, feasible, never worse -- from an immature incumbent, where rounds have many cycle candidates."""
    import kao_cycle as kc
    B, R, P, rf = shape
    pt, t = _drifted(ko, B, R, P, rf)
    for iters in (128, 512, 2048):              # the first feasible incumbent: plenty of improving cycles left
        r = kp.port_search(t, 3, 0, 1, iters)
        if r["best_obj"] >= 0:
            break
    assert r["best_obj"] >= 0
    a = r["best"]
    monkeypatch.setenv("KAO_CX_BULK_SLOTS", "0")
    rounds = 64 if P <= 200 else 4              # (the oracle is pure Python)
    Xo, hist = kc.improve(t, a, rounds, bulk_slots=0)
    Xg, obj, st = kao.improve_cycles(pt, a, rounds)
    assert np.array_equal(np.asarray(Xo).reshape(-1), Xg.reshape(-1))
    assert any(h.get("bulk") for h in hist)
    o, v = ko.verify(t, Xg)
    assert int(np.asarray(v).sum()) == 0 and o == obj >= st["objective_before"]


def test_fixpoint_on_a_1000_partition_topic_is_feasible_and_better(kao, ko, kp):
    pt, t = _drifted(ko, 100, 5, 1000)
    a = kp.port_search(t, 3, 0, 30, 512)["best"]
    X, obj, st = kao.improve_cycles(pt, a, 0)
    o, v = ko.verify(t, X)
    assert int(np.asarray(v).sum()) == 0 and o == obj
    assert obj > st["objective_before"] and st["improving_rounds"] >= 2
    X2, obj2, st2 = kao.improve_cycles(pt, X, 0)           # a fixpoint stays a fixpoint
    assert obj2 == obj and st2["improving_rounds"] == 0 and np.array_equal(X, X2)
    assert obj <= load_golden("drift_scale.json")["rows"][0]["milp_objective"]


def test_high_rf_and_broker_weights_match_the_oracle(kao, ko, kp):
    """Round 3: KAO-CX for RF 5..8 (up to 1,863 seed configurations per partition) and for topics with broker weights (they enter
    every edge and seed price): matrices, midpoints, seed tables, whole rounds and fixpoints bit-exact against the oracle."""
    import kao_cycle as kc
    rng = np.random.default_rng(77)
    n_rf = n_w = n_improved = 0
    for s in range(200):
        cases = []
        t = ko.random_case_rf(s)
        if t.rf > 4 and n_rf < 12:
            cases.append(("rf", t))
        t = ko.random_case_wide(s)
        if 2 <= t.rf <= 4 and n_w < 12:
            t.broker_w = rng.integers(0, 6, t.n_brokers).astype(np.int32)
            t.broker_wl = rng.integers(0, 4, t.n_brokers).astype(np.int32) if s % 3 else None
            cases.append(("w", t))
        for kind, t in cases:
            r = kp.port_search(t, 3, 0, 2, 64)
            if r["best_obj"] < 0:
                continue
            a = r["best"]
            pt = to_product_topic(t)
            rd = _same_matrices(kao, kc, pt, t, a)
            if not rd.cycle_candidates():
                assert np.array_equal(kao.cycle_seeds(pt, a), rd.seed_table()), (kind, s)
            Xo, hist = kc.improve(t, a, 64)
            Xg, obj, st = kao.improve_cycles(pt, a, 64)
            assert np.array_equal(np.asarray(Xo).reshape(-1), Xg.reshape(-1)), (kind, s)
            o, v = ko.verify(t, Xg)
            assert int(np.asarray(v).sum()) == 0 and o == obj == st["objective_after"]
            n_improved += obj > st["objective_before"]
            if kind == "rf":
                n_rf += 1
            else:
                n_w += 1
        if n_rf >= 12 and n_w >= 12:
            break
    assert n_rf >= 8 and n_w >= 8 and n_improved >= 3, (n_rf, n_w, n_improved)


def test_compound_edge_layer_lifts_the_committed_fixpoint(kao, ko, monkeypatch):
    """The next KAO-CX layer (test hook KAO_CX_PAIRS=1; specification oracle/kao_cycle_pairs.py): leader-balanced pairs of leader
    transfers as compound edges of the F graph.  The committed fixpoint of the drifted 300 x 2000 topic (14825, a fixpoint of the
    plain layers, one unit below the MILP optimum) is lifted to 14826 -- as the oracle prototype does -- and stays feasible."""
    import os
    pt, t = _drifted(ko, 300, 6, 2000)
    X = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kao_cx_fixpoint_300x2000_d1.npy")).reshape(2000, 3)
    monkeypatch.delenv("KAO_CX_PAIRS", raising=False)
    X0, obj0, st0 = kao.improve_cycles(pt, X, 0)
    assert obj0 == st0["objective_before"] == 14825 and np.array_equal(X0.reshape(-1), X.reshape(-1))
    monkeypatch.setenv("KAO_CX_PAIRS", "1")
    X1, obj1, st1 = kao.improve_cycles(pt, X, 0)
    o, v = ko.verify(t, X1)
    assert int(np.asarray(v).sum()) == 0 and o == obj1 == 14826 and st1["improving_rounds"] >= 1


def test_rejects_what_it_does_not_support(kao, ko, kp):
    t = next(x for x in (ko.random_case_wide(s) for s in range(400)) if x.rf == 1)   # RF 1: no follower to move, no role to swap
    pt = to_product_topic(t)
    a = np.zeros((t.n_partitions, t.rf), dtype=np.uint16)
    with pytest.raises(kao.KaoError):
        kao.improve_cycles(pt, a, 1)
    pt2, t2 = _drifted(ko, 30, 3, 60)
    bad = np.zeros((60, 3), dtype=np.uint16)               # every replica on broker 0: infeasible start
    with pytest.raises(kao.KaoError):
        kao.improve_cycles(pt2, bad, 1)


def test_solve_with_cycles_proves_the_drifted_1000_partition_topic(kao, ko):
    pt, t = _drifted(ko, 100, 5, 1000)
    want = load_golden("drift_scale.json")["rows"][0]["milp_objective"]
    for cx in (0, -1):
        r = kao.solve([pt], seed=3, stop_at_bound=1, time_limit_s=5.0, use_cycles=cx)[0]
        o, v = ko.verify(t, r.assignment)
        assert int(np.asarray(v).sum()) == 0 and o == r.objective
        if cx == 0:
            assert r.objective == want == r.upper_bound


def test_solve_with_cycles_beats_plain_search_on_a_large_drifted_topic(kao, ko, monkeypatch):
    """(The search engines alone, KAO_LP_ROUND=0: since round 5 the rounded iterate of KAO-LP proves this topic in 0.25 s before
    either of them matters -- tests/test_gpu_lp.py.)"""
    monkeypatch.setenv("KAO_LP_ROUND", "0")
    pt, t = _drifted(ko, 500, 10, 10000)
    with_cx = kao.solve([pt], seed=3, stop_at_bound=1, time_limit_s=2.0)[0]
    without = kao.solve([pt], seed=3, stop_at_bound=1, time_limit_s=2.0, use_cycles=-1)[0]
    o, v = ko.verify(t, with_cx.assignment)
    assert int(np.asarray(v).sum()) == 0 and o == with_cx.objective
    assert with_cx.objective > without.objective
    assert with_cx.upper_bound >= with_cx.objective


@pytest.mark.parametrize("name", ["cfg2_drift.json", "cfg3_drift.json", "cfg4_drift.json"])
def test_solve_with_eager_cycles_keeps_the_golden_optima(kao, ko, name, monkeypatch):
    """KAO-CX forced after EVERY launch (test hook KAO_CX_EAGER=1; normally only stalled topics get it): incumbents fetched from
    the session, improved, adopted back as external elites, over and over on many topics at once -- the proven HiGHS optima
    of the drifted configs must come out exactly as without it."""
    monkeypatch.setenv("KAO_CX_EAGER", "1")
    g = load_golden(name)["topics"]
    ots = [ko.topic_from_dict(e["topic"]) for e in g]
    res = kao.solve([to_product_topic(t) for t in ots], seed=9, time_limit_s=30.0, stop_at_bound=1)
    for e, ot, r in zip(g, ots, res):
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective == e["objective"], (ot.name, r.objective, e["objective"])
        assert r.status == "OPTIMAL_PROVEN" and r.upper_bound == e["objective"]


def test_solve_with_eager_cycles_on_the_wide_family(kao, ko, monkeypatch):
    """The 400-instance wide family (RF 1..4 incl. RF changes, empty current slots, uneven racks, odd weights) in one call with
    KAO-CX after every launch: unsupported topics are skipped, every answer is feasible and never above the HiGHS optimum,
    and the optimum is still reached and proven on all but a few."""
    monkeypatch.setenv("KAO_CX_EAGER", "1")
    cases = load_golden("random_wide.json")["cases"]
    ots = [ko.random_case_wide(c["seed"]) for c in cases]
    res = kao.solve([to_product_topic(t) for t in ots], seed=31, restarts=32, iters_per_launch=256, time_limit_s=30.0, stop_at_bound=1)
    n_opt = n_equal = n_proven = 0
    for c, ot, r in zip(cases, ots, res):
        if c["status"] == "infeasible":
            assert r.status == "INFEASIBLE_PROVEN", c["seed"]
            continue
        n_opt += 1
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective <= c["objective"], c["seed"]
        n_equal += r.objective == c["objective"]
        n_proven += r.status == "OPTIMAL_PROVEN"
    assert n_equal == n_opt == n_proven, (n_opt, n_equal, n_proven)   # 175 / 175 / 175 since round 4 (tools/tol_probe.py); an equality since round 5
