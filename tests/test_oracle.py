"""CPU tests of the oracle itself (no GPU): the restatement is pinned against the README worked
example (KAT-1, the only result the reference pins) and cross-checked three ways -- HiGHS on the
materialised model, HiGHS on the emitted lp_solve text, brute force on tiny instances -- and the C
port is checked against the numpy verifier and against the exact optimum."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden, random_candidates


def test_kat1_readme_example(ko):
    """README.md:52-63 -> README.md:85-91: only partition 1 changes, [8,19] -> [8,1]."""
    t = ko.readme_example()
    assert (t.n_brokers, t.n_racks, t.n_partitions, t.rf) == (19, 2, 10, 2)
    bd = t.bounds()
    assert (bd["rep_lo"], bd["rep_hi"]) == (1, 2)      # README.md:159-160
    assert (bd["lead_lo"], bd["lead_hi"]) == (0, 1)    # README.md:164-165
    assert (bd["rack_lo"], bd["rack_hi"]) == (10, 10)  # README.md:174-175
    assert bd["prack_hi"] == 1                         # README.md:179
    ex = ko.solve_exact(t)
    assert ex.status == "optimal" and ex.objective == 58
    assert ko.count_moves(t, ex.assign) == (1, 0)
    canon = ko.canonicalize(t, ex.assign)
    doc = ko.assignment_to_json([t], [canon])
    want = {p["partition"]: p["replicas"] for p in ko.README_CURRENT["partitions"]}
    want[1] = [8, 1]  # README.md:88
    assert {p["partition"]: p["replicas"] for p in doc["partitions"]} == want
    g = load_golden("kat1.json")
    assert g["objective"] == 58 and g["expected_assignment"] == canon.tolist()


def test_kat1_alt_weights_and_cooptimal(ko):
    t = ko.readme_example()
    t.weights = ((4, 2), (2, 1))
    assert ko.solve_exact(t).objective == 49
    t = ko.readme_example()
    # forcing p1's follower onto each broker: optimal exactly for the odd brokers 1..17 (9 co-optima)
    good = []
    for b in range(19):
        a = np.array(t.current, dtype=np.int64)
        a[1, 1] = b
        obj, viol = ko.verify(t, a)
        if viol[0] == 0 and obj == 58:
            good.append(b)
    assert good == [1, 3, 5, 7, 9, 11, 13, 15, 17]


def test_kafka_tool_proposal_is_worse(ko):
    """README.md:67-78: kafka-reassign-partitions' own proposal moves 20/20 replicas, 10/10 leaders
    and puts partition 2 ([16,0]) in a single AZ."""
    t = ko.readme_example()
    prop = [[14, 17], [15, 18], [16, 0], [17, 1], [18, 2], [0, 3], [1, 4], [2, 5], [3, 6], [4, 7]]
    a = np.array(prop)
    assert ko.count_moves(t, a) == (20, 10)
    obj, viol = ko.verify(t, a)
    assert obj == 0 and viol[7] > 0  # C7: two replicas of p2 in rack a, none in b


def test_lp_text_roundtrip(ko):
    """The emitted lp_solve LP text (README.md:144-185) encodes the same model."""
    t = ko.readme_example()
    txt = ko.write_lp(t)
    assert txt.startswith("// Optimization function") and "\nbin\n" in txt
    first_bin = txt.split("\nbin\n")[1].split(",")[:4]
    assert [s.strip() for s in first_bin] == ["t1b0p0", "t1b0p0_l", "t1b0p1", "t1b0p1_l"]  # README.md:184 order
    status, obj, vals = ko.solve_lp_text(txt)
    assert status == "optimal" and obj == 58
    assert vals["t1b8p1_l"] == 1 and vals["t1b7p0_l"] == 1 and vals["t1b18p0"] == 1
    assert sum(vals.values()) == 20
    rows = ko.build_rows(t)
    assert len(rows) == 2 * 10 + 2 * 19 + 19 * 10 + 2 + 10 * 2  # ranged-row count


@pytest.mark.parametrize("seed", range(6))
def test_bruteforce_matches_highs(ko, seed):
    t = ko.random_case(500 + seed, max_b=5, max_p=3)
    if t.n_brokers > 6 or t.n_partitions > 3:
        pytest.skip("too large for brute force")
    best, _ = ko.brute_force(t)
    ex = ko.solve_exact(t)
    if best is None:
        assert ex.status == "infeasible"
    else:
        assert ex.status == "optimal" and ex.objective == best


def test_port_eval_matches_verifier(ko, kp):
    for t in [ko.readme_example(), ko.gen_config(2).topics[0], ko.gen_config(4, n_topics=1).topics[0],
              ko.random_case(3), ko.random_case(11)]:
        for a in random_candidates(t, 12, 5):
            o1, v1 = ko.verify(t, a)
            o2, v2 = kp.port_eval(t, a)
            assert o1 == o2 and v1.tolist() == v2.tolist()


def test_golden_eval_vectors(ko, kp):
    g = load_golden("kat1.json")
    t = ko.topic_from_dict(g["topic"])
    for e in g["eval_vectors"]:
        a = np.array(e["assignment"], dtype=np.uint16)
        o, v = kp.port_eval(t, a)
        assert o == e["objective"] and v.tolist() == e["viol"]
        o, v = ko.verify(t, a)
        assert o == e["objective"] and v.tolist() == e["viol"]


def test_port_search_reaches_golden_optimum(ko, kp):
    """The scalar replay of the device search finds the exact optimum on the golden instances."""
    cases = load_golden("random_small.json")["cases"] + load_golden("random_medium.json")["cases"][:40]
    n = 0
    for c in cases:
        t = ko.topic_from_dict(c["topic"])
        best = max(kp.port_search(t, 7, rho, 1, 2048)["best_obj"] for rho in range(8))
        if c["status"] != "optimal":
            assert best == -1, c["seed"]  # infeasible instance: no feasible state is ever reported
            continue
        assert best == c["objective"], c["seed"]
        n += 1
    assert n >= 50
    for name in ("cfg3.json", "cfg4.json"):
        e = load_golden(name)["topics"][0]
        t = ko.topic_from_dict(e["topic"])
        best = max(kp.port_search(t, 3, rho, 2, 1024)["best_obj"] for rho in range(4))
        assert best == e["objective"]


def test_port_search_deterministic_and_sound(ko, kp):
    t = ko.gen_config(4, n_topics=1).topics[0]
    r1 = kp.port_search(t, 42, 5, 3, 200)
    r2 = kp.port_search(t, 42, 5, 3, 200)
    assert r1["final"].tolist() == r2["final"].tolist() and r1["best_obj"] == r2["best_obj"]
    r3 = kp.port_search(t, 43, 5, 3, 200)
    assert r3["final"].tolist() != r1["final"].tolist()
    obj, viol = ko.verify(t, r1["final"])
    assert (obj, int(viol[0])) == (r1["obj"], r1["V"])  # incremental bookkeeping == full evaluation
    if r1["best_obj"] >= 0:
        obj, viol = ko.verify(t, r1["best"])
        assert viol[0] == 0 and obj == r1["best_obj"]


def test_infeasible_instance_detected(ko, kp):
    """rigid floor/ceil bands can be jointly infeasible; the exact oracle says so and
    the search reports no feasible state."""
    c = ko.make_cluster("h5", 100, 4, 1, 64, 3, [3, 17, 42, 77, 99], [])
    t = c.topics[0]
    assert ko.solve_exact(t, 60).status == "infeasible"
    r = kp.port_search(t, 1, 0, 1, 2000)
    assert r["best_obj"] == -1


def test_upper_bound_is_valid_and_often_tight(ko):
    """upper_bound_forced (restated by the library's kao_upper_bound) never undercuts the exact optimum and
    closes the gap on BASELINE config 3, where the simple bound is loose (2 forced moves per topic)."""
    n = tight = 0
    for name in ("cfg2.json", "cfg3.json", "cfg4.json"):
        for e in load_golden(name)["topics"]:
            t = ko.topic_from_dict(e["topic"])
            ub = ko.upper_bound_forced(t)
            assert e["objective"] <= ub <= ko.upper_bound_simple(t)
            if name == "cfg3.json":
                assert ub == e["objective"] < ko.upper_bound_simple(t)
    for c in load_golden("random_small.json")["cases"]:
        if c["status"] != "optimal":
            continue
        t = ko.topic_from_dict(c["topic"])
        ub = ko.upper_bound_forced(t)
        assert c["objective"] <= ub <= ko.upper_bound_simple(t), c["seed"]
        n += 1
        tight += ub == c["objective"]
    assert tight >= n // 2


def test_port_bookkeeping_matches_verifier_on_edge_shapes(ko, kp):
    """The replay's incremental (objective, violation) equals the independent numpy verifier on structured
    edge shapes: single-broker racks (rack stride padding), one rack, RF 1, RF = B - 1, tiny P."""
    import itertools
    n = 0
    for B0, R, P, rf in itertools.product((2, 3, 4, 5, 7), (1, 2, 3, 4, 5), (1, 2, 5), (1, 2, 3)):
        if rf >= B0 or R > B0:
            continue
        t = ko.make_cluster("e", B0, R, 1, P, rf, [], []).topics[0]
        r = kp.port_search(t, 3, 0, 1, 32)
        obj, viol = ko.verify(t, r["final"])
        assert (obj, int(viol[0])) == (r["obj"], r["V"]), (B0, R, P, rf)
        n += 1
    assert n > 80
    full = ko.make_cluster("full", 3, 3, 1, 5, 3, [], []).topics[0]  # RF == B, every rack a single broker
    r = kp.port_search(full, 1, 0, 1, 16)
    assert (r["best_obj"], r["V"]) == (40, 0)


def test_port_dual_bound_is_valid_and_closes_the_gap(ko, kp):
    """KAO-DB (oracle/kao_port.c::kao_port_dual_bound, the replay of K-bound): on every feasible instance of the wide
    golden family the Lagrangian dual value never undercuts the HiGHS optimum -- also when the step aims at a
    suboptimal incumbent -- and with the optimum as target it closes the gap (floor(dual) == optimum) on all but a
    few; the closed-form bound alone is tight on fewer than half of them."""
    cases = [c for c in load_golden("random_wide.json")["cases"] if c["status"] == "optimal"]
    closed = tight_closed_form = 0
    for c in cases:
        t = ko.random_case_wide(c["seed"])
        st = kp.port_dual_bound(t, c["objective"], 1500)
        assert not st.flags & 4
        assert st.bound >= c["objective"], c["seed"]
        closed += st.bound == c["objective"]
        tight_closed_form += c["upper_bound"] == c["objective"]
        if c["seed"] % 7 == 0:  # a suboptimal incumbent as target: still a valid bound; a continued run stays one
            lo = kp.port_dual_bound(t, max(0, c["objective"] - 5), 60)
            assert lo.bound >= c["objective"]
            two = kp.port_dual_bound(t, c["objective"], 30)
            if not two.flags & 3:  # (a rounding probe at the end of the short launch may already have closed the gap)
                rec = two.best_L
                two = kp.port_dual_bound(t, c["objective"], 50, two)
                # (round 3: every launch starts with the exact line search along the common shifts, so a split run is no longer
                #  the same trajectory as one long launch; the record only ever improves and stays a valid bound)
                assert two.best_L <= rec and two.bound >= c["objective"] and two.iters <= 80
    assert closed >= len(cases) - 2 and tight_closed_form < len(cases) // 2 + 5, (closed, tight_closed_form, len(cases))
    # an incumbent BELOW the optimum as target (what K-search hands over on hard instances): the level control keeps the
    # certificate close to the optimum anyway (a plain Polyak step stalls 6.5 units above it on average, 25 at worst)
    excess = []
    for c in cases[::3]:
        t = ko.random_case_wide(c["seed"])
        st = kp.port_dual_bound(t, max(0, c["objective"] - max(3, c["objective"] // 50)), 2000)
        assert st.bound >= c["objective"]
        excess.append(st.bound - c["objective"])
    assert sum(excess) <= len(excess) // 4 and max(excess) <= 2, (sum(excess), max(excess))


def test_port_dual_bound_follows_a_moving_incumbent(ko, kp):
    """What kao_solve does to K-bound on a large topic: launches of 150 iterations aimed at incumbents that start far below
    the optimum and close in.  The level control must neither collapse on the way (the first rule halved the distance record ->
    level whenever a stage gained less than half a unit: below a distance of ~2.5 that is every stage) nor be starved by the
    rounding probes; on the drifted 120 x 1200 topic (optimum 8834, proven on the device and equal to the LP value) the
    certificate reaches the optimum although no target ever does."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_drift_scale import oracle_topic
    ot = oracle_topic(120, 4, 1200)
    opt = 8834
    st = None
    for upto, target in ((600, opt - 300), (1200, opt - 50), (1800, opt - 8), (4200, opt - 2)):
        while (st.iters if st else 0) < upto:
            st = kp.port_dual_bound(ot, target, 150, st)
            assert not st.flags & 6
    assert st.bound == opt, (st.bound, st.best_L / kp.DB_SCALE)


def test_port_dual_bound_golden_configs(ko, kp):
    for name in ("kat1.json", "cfg2.json", "cfg3.json", "cfg4.json"):
        g = load_golden(name)
        for e in (g["topics"] if "topics" in g else [g]):
            t = ko.topic_from_dict(e["topic"])
            st = kp.port_dual_bound(t, e["objective"], 400)
            assert st.bound == e["objective"] and st.flags & 1


def test_dual_subproblem_picks_lie_in_the_candidate_pools(ko, kp):
    """The lemma K-bound's kernel relies on (k_bound): under "largest priced value, ties -> lowest
    broker index" every follower the brute-force subproblem picks is one of the partition's current brokers or one of the
    RF best brokers (by generic value F, then index) of one of the RF best racks; the leader is a set member, a current
    broker, one of the RF+1 best brokers (by generic leader value FL) of a rack holding a set member, or the best broker of
    one of the RF+1 best racks by FL.  Checked against the brute-force scan with random multipliers, heavy on ties."""
    rng = np.random.default_rng(7)
    checked = 0
    for seed in range(0, 120, 3):
        t = ko.random_case_wide(seed)
        if ko.provably_infeasible(t):
            continue
        B, R, RF = t.n_brokers, t.n_racks, t.rf
        rack = t.rack_of.astype(int)
        for trial in range(3):
            step = [1, kp.DB_SCALE // 8, kp.DB_SCALE][trial]          # coarse multipliers -> many exact ties
            a = (rng.integers(-3, 4, B) * step).astype(np.int32)
            l = (rng.integers(-2, 3, B) * step).astype(np.int32)
            g = (rng.integers(-2, 3, max(1, R)) * step).astype(np.int32)
            F = -a.astype(np.int64) - g[rack]
            FL = F - l
            topF = {r: sorted(np.flatnonzero(rack == r), key=lambda b: (-F[b], b))[:RF] for r in range(R)}
            topL = {r: sorted(np.flatnonzero(rack == r), key=lambda b: (-FL[b], b))[:RF + 1] for r in range(R)}
            racksF = sorted((r for r in range(R) if topF[r]), key=lambda r: (-F[topF[r][0]], topF[r][0]))[:RF]
            racksL = sorted((r for r in range(R) if topL[r]), key=lambda r: (-FL[topL[r][0]], topL[r][0]))[:RF + 1]
            poolF = {int(b) for r in racksF for b in topF[r]}
            for p in range(0, t.n_partitions, max(1, t.n_partitions // 12)):
                got = kp.port_dual_partition(t, p, a, l, g)
                if got is None:
                    continue
                S, G, _ = got
                cur = {int(b) for b in t.current[p] if b < B}
                for b in G:       # every greedy pick: a current broker or in the follower pool
                    assert b in cur or b in poolF, (seed, trial, p, b)
                poolL = set(G) | cur | {int(topL[r][0]) for r in racksL}
                for b in G:       # RF+1 best leaders of every rack that holds a set member
                    poolL |= {int(x) for x in topL[rack[b]]}
                assert S[0] in poolL, (seed, trial, p, S[0])
                assert set(S[1:]) <= set(G)
                checked += 1
    assert checked > 300


def test_dual_value_is_an_upper_bound_for_any_multipliers(ko, kp):
    """Weak duality, the claim the certificate rests on: for ARBITRARY multipliers (not ones the iteration produced) the
    dual value L(a,l,g) = sum of the exact partition subproblems + multiplier x priced band end is >= the HiGHS optimum."""
    rng = np.random.default_rng(11)
    cases = [c for c in load_golden("random_wide.json")["cases"] if c["status"] == "optimal"][::5]
    cases += [c for c in load_golden("random_small.json")["cases"] if c["status"] == "optimal"][::4]
    n = 0
    for c in cases:
        t = ko.random_case_wide(c["seed"]) if "shape" in c else ko.topic_from_dict(c["topic"])
        bd = t.bounds()
        B, R = t.n_brokers, t.n_racks
        for scale in (0, 300, kp.DB_SCALE, 5 * kp.DB_SCALE):
            a = rng.integers(-scale, scale + 1, B).astype(np.int32)
            l = rng.integers(-scale, scale + 1, B).astype(np.int32)
            g = rng.integers(-scale, scale + 1, max(1, R)).astype(np.int32)
            L = 0
            for p in range(t.n_partitions):
                S, G, v = kp.port_dual_partition(t, p, a, l, g)
                L += v
            L += int(np.where(a > 0, a.astype(np.int64) * bd["rep_hi"], a.astype(np.int64) * bd["rep_lo"]).sum())
            L += int(np.where(l > 0, l.astype(np.int64) * bd["lead_hi"], l.astype(np.int64) * bd["lead_lo"]).sum())
            L += int(np.where(g[:R] > 0, g[:R].astype(np.int64) * bd["rack_hi"], g[:R].astype(np.int64) * bd["rack_lo"]).sum())
            assert L >= c["objective"] * kp.DB_SCALE, (c["seed"], scale, L / kp.DB_SCALE, c["objective"])
            n += 1
    assert n >= 150


def test_port_generations_differ_and_are_reproducible(ko, kp):
    """The scalar restatement of a new generation (kao_port_run_launch with gen > 0): the restart is re-initialised with the
    generation number in the tie-break hash -- reproducibly, and differently from generation 0 and from generation 2."""
    ot = ko.gen_config(2).topics[0]

    def run(gens):
        r = kp.PortRun(ot, 0xABCDEF, 5)
        for launch, g in enumerate(gens):
            r.launch(launch, 60, gen=g)
        out = r.read()
        r.close()
        return out
    a, b = run([0, 0, 1, 0]), run([0, 0, 1, 0])
    assert np.array_equal(a["final"], b["final"]) and a["obj"] == b["obj"]
    c, d = run([0, 0, 0, 0]), run([0, 0, 2, 0])
    assert not np.array_equal(a["final"], c["final"]) and not np.array_equal(a["final"], d["final"])
    obj, viol = ko.verify(ot, a["final"])
    assert obj == a["obj"] and int(viol[0]) == a["V"]
