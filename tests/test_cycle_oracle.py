"""KAO-CX oracle (oracle/kao_cycle.py) on the CPU: the closures are shortest paths, a realised candidate changes the objective
by exactly the value it was priced at, rounds only ever improve a feasible assignment, and the result never exceeds the exact
optimum (HiGHS) of the golden instances."""
import numpy as np
import pytest

from conftest import load_golden


def _incumbent(kp, t, launches=6, iters=256, rho=0):
    r = kp.port_search(t, 3, rho, launches, iters)
    return r["best"] if r["best_obj"] >= 0 else None


def _wide_cases(ko, kp, seeds, need_rf=2, launches=6, iters=256):
    for s in seeds:
        t = ko.random_case_wide(s)
        if t.rf < need_rf or t.rf > 4:
            continue
        a = _incumbent(kp, t, launches, iters)
        if a is not None:
            yield s, t, a


def test_closure_is_the_bounded_shortest_path(ko, kp):
    import kao_cycle as kc
    n_checked = 0
    for s, t, a in _wide_cases(ko, kp, range(40)):
        rd = kc.Round(t, a)
        for D, M in ((rd.DF, rd.MF), (rd.DS, rd.MS), (rd.DL, rd.ML)):
            ref = D[0].copy()
            for _ in range(7):      # paths of <= 8 edges: seven more relaxations of the edge matrix
                ref = np.minimum(ref, (ref[:, :, None] + D[0][None, :, :]).min(axis=1))
            assert np.array_equal(np.minimum(ref, kc.CINF), D[3]), s
            # the midpoints unroll into a walk of <= 8 edges whose edge costs add up to the entry (paths are only unrolled
            # when no level has a negative cycle: otherwise the cycles themselves are the candidates)
            if any((np.diag(D[lev]) < 0).any() for lev in range(1, 4)):
                continue
            n = D[0].shape[0]
            for u in range(0, n, 3):
                for v in range(0, n, 5):
                    if D[3][u, v] >= kc.CINF or u == v:
                        continue
                    pth = rd._path(M, u, v, 3)
                    assert pth[0] == u and pth[-1] == v and len(pth) - 1 <= 8
                    assert sum(int(D[0][x, y]) for x, y in zip(pth[:-1], pth[1:]) if x != y) == D[3][u, v]
        n_checked += 1
    assert n_checked >= 10


def test_realised_candidates_are_worth_their_price(ko, kp):
    import kao_cycle as kc
    seen_seed = seen_cycle = 0
    import itertools
    # mature incumbents (after the two-slot REPLACE scan of round 4 most of them are fixpoints already) and immature ones
    for s, t, a in itertools.chain(_wide_cases(ko, kp, range(60)), _wide_cases(ko, kp, range(60), launches=2, iters=64)):
        rd = kc.Round(t, a)
        base, v0 = kc.evaluate(t, rd.A)
        assert v0 == 0
        cyc = rd.cycle_candidates()
        if cyc:
            for c in cyc[:20]:
                for X, used in rd.realise_cycle(c):
                    o, _ = kc.evaluate(t, X)
                    assert o - base == c[0], (s, c)
                    seen_cycle += 1
        else:
            for c in rd.seed_candidates()[:40]:
                for X, used in rd.realise_seed(c):
                    o, _ = kc.evaluate(t, X)
                    assert o - base == c[0], (s, c)
                    seen_seed += 1
    assert seen_seed > 0 and seen_cycle > 0


def test_rounds_only_improve_and_stay_feasible(ko, kp):
    import kao_cycle as kc
    improved = 0
    for s, t, a in _wide_cases(ko, kp, range(100, 160), launches=1, iters=48):
        base, _ = kc.evaluate(t, a)
        X, hist = kc.improve(t, a, 8)
        o, v = kc.evaluate(t, X)
        assert v == 0 and o >= base
        objs = [h["objective"] for h in hist if "objective" in h]
        assert objs == sorted(set(objs)) and (not objs or (objs[0] > base and objs[-1] == o))
        improved += o > base
    assert improved >= 3


def test_never_above_the_exact_optimum(ko, kp):
    import kao_cycle as kc
    g = load_golden("random_wide.json")
    n = 0
    for case in g["cases"][:120]:
        if case["status"] != "optimal":
            continue
        t = ko.random_case_wide(case["seed"])
        if t.rf < 2 or t.rf > 4:
            continue
        a = _incumbent(kp, t, launches=2, iters=64)
        if a is None:
            continue
        X, _ = kc.improve(t, a, 6)
        o, v = kc.evaluate(t, X)
        assert v == 0 and o <= case["objective"]
        n += 1
    assert n >= 10


def test_config_numbering_round_trips(ko, kp):
    import kao_cycle as kc
    for s, t, a in _wide_cases(ko, kp, range(12)):
        rd = kc.Round(t, a)
        assert rd.seed_table().shape[1] == kc.n_cfg(t.rf, t.rf_cur)
        for tot, p, cfg, y in rd.seed_candidates()[:50]:
            row = rd.seed_row(p, cfg, y % 4096)
            assert len(set(row)) == t.rf and max(row) < t.n_brokers


def _rf_and_weighted_cases(ko, kp, n_rf=10, n_w=10):
    """RF 5..8 topics (random_case_rf) and wide-family topics carrying broker weights."""
    rng = np.random.default_rng(77)
    got_rf = got_w = 0
    for s in range(200):
        if got_rf < n_rf:
            t = ko.random_case_rf(s)
            if t.rf > 4:
                a = _incumbent(kp, t, 2, 64)
                if a is not None:
                    got_rf += 1
                    yield "rf", s, t, a
        if got_w < n_w:
            t = ko.random_case_wide(s)
            if 2 <= t.rf <= 4:
                t.broker_w = rng.integers(0, 6, t.n_brokers).astype(np.int32)
                t.broker_wl = rng.integers(0, 4, t.n_brokers).astype(np.int32) if s % 3 else None
                a = _incumbent(kp, t, 2, 64)
                if a is not None:
                    got_w += 1
                    yield "w", s, t, a
        if got_rf >= n_rf and got_w >= n_w:
            break


def test_high_rf_and_broker_weights(ko, kp):
    """RF 5..8 and broker weights (round 3): priced candidates are worth exactly their price under the independent verifier,
    rounds only improve, the result stays feasible."""
    import kao_cycle as kc
    seen = {"rf": 0, "w": 0}
    priced = improved = 0
    for kind, s, t, a in _rf_and_weighted_cases(ko, kp):
        assert kc.supported(t)
        rd = kc.Round(t, a)
        base, v0 = kc.evaluate(t, rd.A)
        assert v0 == 0
        cyc = rd.cycle_candidates()
        cands = [("c", c) for c in cyc[:10]] if cyc else [("s", c) for c in rd.seed_candidates()[:20]]
        for k2, c in cands:
            for i, (X, used) in enumerate(rd.realise_cycle(c) if k2 == "c" else rd.realise_seed(c)):
                o, _ = kc.evaluate(t, X)   # the second realisation of a two-replica seed pairs the closures the dearer way round
                assert (o - base == c[0]) if i == 0 else (o - base <= c[0]), (kind, s, c)
                priced += 1
        X, hist = kc.improve(t, a, 8)
        o, v = kc.evaluate(t, X)
        assert v == 0 and o >= base
        improved += o > base
        seen[kind] += 1
    assert seen["rf"] >= 5 and seen["w"] >= 5 and priced > 0 and improved >= 2


@pytest.mark.slow
def test_compound_edges_of_leader_balanced_pairs_improve_a_fixpoint(ko, kp):
    """oracle/kao_cycle_pairs.py (prototype of the next KAO-CX layer; ~2.5 minutes of pure Python, -m slow): a KAO-CX fixpoint of
    the drifted 300 x 2000 topic one unit below the MILP optimum 14826 (tests/golden/kao_cx_fixpoint_300x2000_d1.npy, made by
    the scalar replay of K-search + oracle KAO-CX) is improved to the optimum by a cycle through two compound edges --
    leader-balanced pairs of leader transfers whose net replica effect is one unit."""
    import os
    import kao_cycle as kc
    import kao_cycle_pairs as kcp
    t = kcp.drift_topic(300, 6, 2000, 1)
    X = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kao_cx_fixpoint_300x2000_d1.npy")).reshape(2000, 3)
    base, v0 = kc.evaluate(t, X)
    assert (base, v0) == (14825, 0)
    again, hist = kc.improve(t, X, max_rounds=4)
    assert kc.evaluate(t, again)[0] == 14825           # a fixpoint of plain KAO-CX
    Y, objs = kcp.improve_with_pairs(t, X, max_passes=2, verbose=False)
    obj, viol = ko.verify(t, np.asarray(Y).astype(np.uint16))
    assert int(np.asarray(viol).sum()) == 0 and obj == objs[-1] == 14826
    assert [r for r in load_golden("drift_scale.json")["rows"] if (r["B"], r["P"]) == (300, 2000)][0]["milp_objective"] == 14826


def test_bulk_rounds_only_improve_and_end_no_lower(ko, kp):
    """Round 4: the bulk mode of a round (topics beyond 131,072 replica slots: candidates of every level, a partition-disjoint set
    merged before scoring), forced onto small topics: every round keeps the assignment feasible and strictly improves it, and bulk
    rounds do occur from immature incumbents (the one-by-one mode runs beside it from the same start: both end feasible)."""
    import kao_cycle as kc
    n = n_bulk = 0
    for s, t, a in _wide_cases(ko, kp, range(60), launches=2, iters=64):
        base, v0 = kc.evaluate(t, np.asarray(a).reshape(t.n_partitions, t.rf))
        assert v0 == 0
        Xb, hist_b = kc.improve(t, a, 64, bulk_slots=0)
        Xo, hist_o = kc.improve(t, a, 64)
        prev = base
        for h in hist_b:
            if h.get("objective") is not None:
                assert h["objective"] > prev, s
                prev = h["objective"]
            n_bulk += bool(h.get("bulk"))
        ob, vb = kc.evaluate(t, Xb)
        oo, vo = kc.evaluate(t, Xo)
        assert vb == 0 and vo == 0 and ob == prev >= base, s
        n += 1
        if n >= 16:
            break
    assert n >= 10 and n_bulk >= 3, (n, n_bulk)
