"""GPU parity tests (run on the MI355X box with -m gpu).  Everything goes through the C ABI of
libkao.so; the oracle (numpy verifier, C port, HiGHS golden optima) is only the checker.

  * K-eval  vs oracle verifier: bit-exact (objective, viol[8]) -- golden vectors, seeded random
    candidates at every BASELINE config's topic size, edge cases (empty slots, out-of-range ids,
    duplicates, RF change).
  * K-search vs the scalar replay (oracle/kao_port.c): bit-identical final state, best snapshot and
    counters per restart, for the same seed.
  * K-search vs exact optimum (HiGHS golden): equal objective; README KAT-1 reproduced bit-exactly
    after the canonical tie-break; unique optima reproduced bit-exactly.
  * size-independent properties at full BASELINE sizes: every returned assignment feasible under the
    independent verifier, objective == verifier objective, objective <= upper bound, idempotence
    (re-solving the solution moves nothing), drift counter == 0.
"""
import numpy as np
import pytest

from conftest import ROOT as ROOT_DIR, load_golden, random_candidates, to_product_topic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kao():
    import kafka_assignment_optimizer_amd as k
    k.init(0)
    assert "gfx950" in k.device_name(), k.device_name()
    return k


# ------------------------------------------------------------------------------- K-eval
def test_eval_golden_vectors(kao, ko):
    g = load_golden("kat1.json")
    ot = ko.topic_from_dict(g["topic"])
    pt = to_product_topic(ot)
    cands = np.array([e["assignment"] for e in g["eval_vectors"]], dtype=np.uint16)
    obj, viol = kao.evaluate_batch(pt, cands)
    assert obj.tolist() == [e["objective"] for e in g["eval_vectors"]]
    assert viol.tolist() == [e["viol"] for e in g["eval_vectors"]]
    o1, v1 = kao.evaluate(pt, np.array(g["expected_assignment"], dtype=np.uint16))
    assert (o1, v1.tolist()) == (58, [0] * 8)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_eval_random_candidates_match_oracle(kao, ko, kp, cfg):
    ot = ko.gen_config(cfg, n_topics=1).topics[0]
    pt = to_product_topic(ot)
    n = 200 if cfg != 2 else 96
    cands = random_candidates(ot, n, seed=100 + cfg)
    obj, viol = kao.evaluate_batch(pt, cands)
    for i in range(n):
        o, v = kp.port_eval(ot, cands[i])
        assert (int(obj[i]), viol[i].tolist()) == (o, v.tolist()), (cfg, i)
    for i in range(0, n, 17):  # and the numpy restatement on a subset
        o, v = ko.verify(ot, cands[i])
        assert (int(obj[i]), viol[i].tolist()) == (o, v.tolist())


def test_eval_edge_cases(kao, ko, kp):
    # RF change, single rack, RF 1, RF 4, ragged sizes (P not a multiple of 64, B < 64 and > 64)
    seeds = [s for s in range(60) if ko.random_case(s).rf <= 4 and ko.random_case(s).rf_cur <= 4][:30]
    for s in seeds:
        ot = ko.random_case(s, max_b=14, max_p=10)
        pt = to_product_topic(ot)
        cands = random_candidates(ot, 9, seed=s, p_mut=0.3, p_none=0.1)
        obj, viol = kao.evaluate_batch(pt, cands)
        for i in range(len(cands)):
            o, v = kp.port_eval(ot, cands[i])
            assert (int(obj[i]), viol[i].tolist()) == (o, v.tolist()), (s, i)
    for R, B0 in ((100, 300), (200, 400)):  # more racks than lanes
        ot = ko.make_cluster("r", B0, R, 1, 77, 3, [5], [(B0, 5)]).topics[0]
        cands = random_candidates(ot, 12, seed=R, p_mut=0.3)
        obj, viol = kao.evaluate_batch(to_product_topic(ot), cands)
        for i in range(len(cands)):
            o, v = kp.port_eval(ot, cands[i])
            assert (int(obj[i]), viol[i].tolist()) == (o, v.tolist()), (R, i)
    c = ko.make_cluster("rf4", 70, 5, 1, 130, 4, [1, 2, 3], [(70, 0), (71, 4)])
    ot = c.topics[0]
    cands = random_candidates(ot, 40, seed=9)
    cands[1][:] = 0xFFFF  # a completely empty candidate
    obj, viol = kao.evaluate_batch(to_product_topic(ot), cands)
    for i in range(len(cands)):
        o, v = kp.port_eval(ot, cands[i])
        assert (int(obj[i]), viol[i].tolist()) == (o, v.tolist())
    assert viol[1][1] == 130 * 4 and viol[1][2] == 130


# ------------------------------------------------------------------------------- K-search replay
@pytest.mark.parametrize("cfg,launches,iters", [(1, 2, 96), (2, 2, 160), (4, 3, 128), (5, 1, 200)])
def test_search_replay_bit_exact(kao, ko, kp, cfg, launches, iters):
    """Same seed -> the device restart and the scalar replay agree bit for bit."""
    ots = ko.gen_config(cfg, n_topics=2).topics
    pts = [to_product_topic(t) for t in ots]
    seed = 0xABCDEF12345 + cfg
    with kao.Session(pts, seed=seed, restarts=8, iters_per_launch=iters) as s:
        s.step(launches)
        s.sync()
        st = s.stats()
        assert st["drift"] == 0
        n_eval = 0
        for ti, ot in enumerate(ots):
            tseed = seed ^ (((ti + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
            for rho in (0, 3, 7):
                dev = s.restart_state(ti, rho)
                ref = kp.port_search(ot, tseed, rho, launches, iters)
                n_eval += ref["n_eval"]  # 3 of the 8 restarts replayed; every restart of a topic counts the same
                assert dev["final"].tolist() == ref["final"].tolist(), (cfg, ti, rho)
                assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == \
                       (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"]), (cfg, ti, rho)
                if ref["best_obj"] >= 0:
                    assert dev["best"].tolist() == ref["best"].tolist()
        assert st["delta_candidates"] * 3 == n_eval * 8  # the host's neighbour count is the replay's count


def _tseed(seed, ti):
    return seed ^ (((ti + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)


def _drifted(ko, cfg, n):
    """First n topics of BASELINE config `cfg` after a 20 % drift (the instances where prices and elites matter)."""
    from kafka_assignment_optimizer_amd import synthetic
    out = []
    for pt in synthetic.drift(synthetic.make_config(cfg, n_topics=n), 0.2, 1):
        out.append(ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                            n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights,
                            bounds_override=dict(pt.bounds_override)))
    return out


@pytest.mark.parametrize("cfg", [2, 4])
def test_search_replay_with_prices_bit_exact(kao, ko, kp, cfg):
    """K-search with Lagrangian prices in the move cost (k_search<*, true>): host-set prices -- random multiples of a
    quarter, and odd values that exercise the rounding to key units -- replayed bit for bit by the port, init included."""
    ots = _drifted(ko, cfg, 2)
    pts = [to_product_topic(t) for t in ots]
    rng = np.random.default_rng(cfg)
    prices = []
    for t in ots:
        a = rng.integers(-8, 9, t.n_brokers) * 16384
        l = rng.integers(-4, 5, t.n_brokers) * 16384 + rng.integers(-4800, 4800, t.n_brokers)   # off-grid on purpose
        g = rng.integers(-2, 3, t.n_racks) * 16384
        prices.append((a.astype(np.int32), l.astype(np.int32), g.astype(np.int32)))
    seed = 0x51CE + cfg
    with kao.Session(pts, seed=seed, restarts=8, iters_per_launch=150) as s:
        for ti, pr in enumerate(prices):
            s.set_prices(ti, *pr)
        s.step(2)
        assert s.stats()["drift"] == 0
        for ti, ot in enumerate(ots):
            for rho in (0, 5):
                run = kp.PortRun(ot, _tseed(seed, ti), rho)
                run.launch(0, 150, prices=prices[ti])
                run.launch(1, 150, prices=prices[ti])
                ref = run.read()
                dev = s.restart_state(ti, rho)
                assert dev["final"].tolist() == ref["final"].tolist(), (cfg, ti, rho)
                assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])
                # prices steer the search but never change what is reported: V / obj are the true values
                obj, viol = ko.verify(ot, dev["final"])
                assert (obj, int(viol[0])) == (dev["obj"], dev["V"])
    # zero prices == the unpriced kernel
    with kao.Session(pts, seed=seed, restarts=8, iters_per_launch=150) as s0, kao.Session(pts, seed=seed, restarts=8, iters_per_launch=150) as s1:
        for ti, t in enumerate(ots):
            s1.set_prices(ti, np.zeros(t.n_brokers, np.int32), np.zeros(t.n_brokers, np.int32), np.zeros(t.n_racks, np.int32))
        s0.step(2); s1.step(2)
        for ti in range(len(ots)):
            assert s0.restart_state(ti, 3)["final"].tolist() == s1.restart_state(ti, 3)["final"].tolist()


def test_adopted_prices_are_the_rounded_multipliers(kao, ko, kp):
    """kao_session_adopt_prices: K-search then carries the multipliers of K-bound's record dual value rounded to the quarter
    grid (checked against the port's replay of that K-bound launch), and the priced launch is replayed by the port."""
    ots = _drifted(ko, 4, 2)
    pts = [to_product_topic(t) for t in ots]
    seed = 99
    with kao.Session(pts, seed=seed, restarts=8, iters_per_launch=100) as s:
        s.step(1)
        res = s.best()
        s.bound_step([max(0, r.objective) for r in res], 60)
        s.bounds()
        s.adopt_prices()
        s.step(1)
        for ti, ot in enumerate(ots):
            st = kp.port_dual_bound(ot, max(0, res[ti].objective), 60)   # the port's replay of the same K-bound launch
            pr = s.prices(ti)
            assert np.any(pr[0] != 0) or np.any(pr[1] != 0)
            # exported = the multipliers of the record dual value, rounded to the quarter grid
            assert (pr[0].tolist(), pr[1].tolist(), pr[2].tolist()) == \
                   (kp.quarter_round(st.ra).tolist(), kp.quarter_round(st.rl).tolist(), kp.quarter_round(st.rg)[:ot.n_racks].tolist())
            run = kp.PortRun(ot, _tseed(seed, ti), 2)
            run.launch(0, 100)
            run.launch(1, 100, prices=pr)
            ref = run.read()
            dev = s.restart_state(ti, 2)
            assert dev["final"].tolist() == ref["final"].tolist(), ti
            assert (dev["best_obj"], dev["V"], dev["obj"]) == (ref["best_obj"], ref["V"], ref["obj"])


def test_elite_launch_replay_bit_exact(kao, ko, kp):
    """Elite launches (every elite_period-th launch trailing restarts may re-seed from the topic's best assignment):
    the port replays the rule from the best assignment the device reported before that launch."""
    ots = _drifted(ko, 4, 3)
    pts = [to_product_topic(t) for t in ots]
    seed = 4242
    iters = 120
    with kao.Session(pts, seed=seed, restarts=16, iters_per_launch=iters, elite_period=2) as s:
        s.step(2)                      # launches 0, 1
        res = s.best()                 # what launch 2 (an elite launch) re-seeds from
        s.step(1)
        assert s.stats()["drift"] == 0
        reseeded = 0
        for ti, ot in enumerate(ots):
            r = res[ti]
            assert r.status != "NO_FEASIBLE"
            for rho in range(16):
                run = kp.PortRun(ot, _tseed(seed, ti), rho)
                run.launch(0, iters)
                run.launch(1, iters)
                before = run.read()
                run.launch(2, iters, elite=(r.assignment, r.objective, r.best_restart))
                ref = run.read()
                dev = s.restart_state(ti, rho)
                assert dev["final"].tolist() == ref["final"].tolist(), (ti, rho)
                assert (dev["best_obj"], dev["V"], dev["obj"]) == (ref["best_obj"], ref["V"], ref["obj"])
                reseeded += before["best_obj"] < r.objective and ref["best_obj"] >= r.objective
        assert reseeded > 0            # the rule fired somewhere


def test_new_generation_replay_bit_exact(kao, ko, kp):
    """kao_session_new_generation (what kao_solve does with a population that converged without a proof): the next launch
    re-initialises every restart with the generation number in the tie-break hash, old snapshots and best keys are dropped, the
    launch counter carries on -- the scalar restatement re-initialises the same way, restart by restart, bit for bit."""
    ot = _drifted(ko, 2, 1)[0]
    seed, iters = 0x6E47, 120
    with kao.Session([to_product_topic(ot)], seed=seed, restarts=8, iters_per_launch=iters) as s:
        s.step(2)
        before = s.best()[0]
        s.new_generation()
        assert int(s.best_keys()[0]) == (1 << 64) - 1                       # nothing of the old generation is left on the device
        s.step(2)
        assert s.stats()["drift"] == 0
        tseed = seed ^ (0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
        states = set()
        for rho in (0, 3, 7):
            dev = s.restart_state(0, rho)
            run = kp.PortRun(ot, tseed, rho)
            run.launch(0, iters); run.launch(1, iters); run.launch(2, iters, gen=1); run.launch(3, iters)
            ref = run.read()
            assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])
            assert np.array_equal(dev["final"], ref["final"]) and np.array_equal(dev["best"], ref["best"])
            plain = kp.port_search(ot, tseed, rho, 4, iters)                 # the same restart without the new generation
            assert not np.array_equal(plain["final"], ref["final"])
            states.add(dev["final"].tobytes())
        assert len(states) == 3 and before.status != "NO_FEASIBLE"


def test_search_replay_random_small(kao, ko, kp):
    cases = [c for c in load_golden("random_small.json")["cases"]][:16]
    ots = [ko.topic_from_dict(c["topic"]) for c in cases]
    seed = 77
    with kao.Session([to_product_topic(t) for t in ots], seed=seed, restarts=4, iters_per_launch=300) as s:
        s.step(2)
        for ti, ot in enumerate(ots):
            tseed = seed ^ (((ti + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
            for rho in range(4):
                dev = s.restart_state(ti, rho)
                ref = kp.port_search(ot, tseed, rho, 2, 300)
                assert dev["final"].tolist() == ref["final"].tolist(), (ti, rho)
                assert (dev["best_obj"], dev["V"], dev["obj"]) == (ref["best_obj"], ref["V"], ref["obj"])
        assert s.stats()["drift"] == 0


@pytest.mark.parametrize("B,R,P", [(500, 10, 5000), (300, 6, 9000)])
def test_search_replay_working_words_in_lds(kao, ko, kp, B, R, P):
    """Round 5 (k_search_curg): between ~4,900 and ~9,800 partitions a restart's WORKING assignment fits LDS alone -- the current
    assignment is read from global memory / L2 -- where rounds 1-4 ran the whole topic from HBM (500 x 5000: 4.8 ms a launch).  Same
    arithmetic: final state, best snapshot, counters and the neighbour count equal the scalar replay bit for bit, without prices and
    with host-set prices; the workgroup's LDS stays within 160 KiB; KAO_CUR_GLOBAL=0 still gives the HBM path (same results)."""
    from kafka_assignment_optimizer_amd import synthetic as sy
    pt = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    ot = ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                  n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights)
    seed = 20250922
    tseed = _tseed(seed, 0)
    rng = np.random.default_rng(7)
    prices = (rng.integers(-8, 9, B).astype(np.int32) * 16384, rng.integers(-4, 5, B).astype(np.int32) * 16384, rng.integers(-4, 5, R).astype(np.int32) * 16384)
    for priced in (False, True):
        with kao.Session([pt], seed=seed, restarts=4, iters_per_launch=200) as s:
            st = s.stats()
            assert 16 * P <= st["lds_bytes_search"] <= 160 * 1024 and st["lds_bytes_search"] < 32 * P    # the working words, not both
            if priced:
                s.set_prices(0, *prices)
            s.step(2)
            assert s.stats()["drift"] == 0
            for rho in (0, 3):
                dev = s.restart_state(0, rho)
                if priced:
                    run = kp.PortRun(ot, tseed, rho)
                    run.launch(0, 200, prices=prices); run.launch(1, 200, prices=prices)
                    ref = run.read(); run.close()
                else:
                    ref = kp.port_search(ot, tseed, rho, 2, 200)
                assert dev["final"].tolist() == ref["final"].tolist(), (priced, rho)
                assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"]), (priced, rho)
                if ref["best_obj"] >= 0:
                    assert dev["best"].tolist() == ref["best"].tolist()


def test_search_replay_varied_shapes(kao, ko, kp):
    """One session holding heterogeneous topics: RF 1/2/4, an RF increase and decrease, a single rack,
    uneven racks (padding slots in the internal index), P not a multiple of 64, B < 64 and B > 64, 100 and 200 racks
    (more racks than wavefront lanes)."""
    mk = ko.make_cluster
    ots = [
        mk("rf1", 10, 2, 1, 7, 1, [3], [(10, 0)]).topics[0],
        mk("rf2", 30, 3, 1, 70, 2, [0, 1, 2, 3], []).topics[0],
        mk("rf4", 70, 5, 1, 130, 4, [1, 2, 3], [(70, 0), (71, 4)]).topics[0],
        mk("rf2to3", 24, 4, 1, 40, 2, [5], [(24, 1), (25, 1), (26, 1)], new_rf=3).topics[0],
        mk("rf3to2", 24, 4, 1, 33, 3, [], [(24, 0)], new_rf=2).topics[0],
        mk("onerack", 9, 1, 1, 12, 3, [4], []).topics[0],
        mk("uneven", 40, 4, 1, 65, 3, [0, 4, 8, 12, 16, 1], []).topics[0],
        mk("racks100", 300, 100, 1, 120, 3, [1, 2, 3], [(300, 1), (301, 2), (302, 3)]).topics[0],   # more racks than lanes
        mk("racks200", 400, 200, 1, 90, 2, [7], [(400, 7)]).topics[0],
    ]
    seed = 424242
    with kao.Session([to_product_topic(t) for t in ots], seed=seed, restarts=8, iters_per_launch=96) as s:
        s.step(3)
        assert s.stats()["drift"] == 0
        for ti, ot in enumerate(ots):
            tseed = seed ^ (((ti + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
            for rho in (0, 1, 2, 5):
                dev = s.restart_state(ti, rho)
                ref = kp.port_search(ot, tseed, rho, 3, 96)
                assert dev["final"].tolist() == ref["final"].tolist(), (ot.name, rho)
                assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == \
                       (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"]), (ot.name, rho)
                if ref["best_obj"] >= 0:
                    assert dev["best"].tolist() == ref["best"].tolist()
        res = s.best()
    for ot, r in zip(ots, res):
        if r.status not in ("NO_FEASIBLE", "INFEASIBLE_PROVEN"):
            obj, viol = ko.verify(ot, r.assignment)
            assert viol[0] == 0 and obj == r.objective, ot.name


def test_heterogeneous_session_uses_launch_groups(kao, ko, kp):
    """A 3000-partition topic next to small ones: topics are bucketed by LDS footprint into separate launches, so the
    small topics keep 4 restarts per workgroup; replay parity holds in every group."""
    big = ko.make_cluster("big3000", 1000, 20, 1, 3000, 3, [7, 77, 777], [(1000, 7), (1001, 17), (1002, 17)]).topics[0]
    small = ko.gen_config(4, n_topics=6).topics
    mid = ko.gen_config(2).topics[0]
    ots = [small[0], big, small[1], mid] + small[2:]
    seed = 555
    with kao.Session([to_product_topic(t) for t in ots], seed=seed, restarts=8, iters_per_launch=64) as s:
        st = s.stats()
        assert st["launch_groups"] >= 2
        # 6 small + 1 mid topics at 4 waves per workgroup (2 workgroups each), the big one at 2 waves (4 workgroups)
        assert st["blocks_search"] == 7 * 2 + 4
        s.step(2)
        assert s.stats()["drift"] == 0
        for ti, ot in enumerate(ots):
            tseed = seed ^ (((ti + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
            dev = s.restart_state(ti, 5)
            ref = kp.port_search(ot, tseed, 5, 2, 64)
            assert dev["final"].tolist() == ref["final"].tolist(), ot.name
            assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])
        for ot, r in zip(ots, s.best()):
            if r.status not in ("NO_FEASIBLE", "INFEASIBLE_PROVEN"):
                obj, viol = kp.port_eval(ot, r.assignment)
                assert viol[0] == 0 and obj == r.objective


def test_empty_and_degenerate_inputs(kao, ko):
    import ctypes as C
    from kafka_assignment_optimizer_amd import _ffi
    lib = _ffi.load()
    h = C.c_void_p()
    assert lib.kao_session_create(None, 0, None, C.byref(h)) == -1          # no topics
    pt = to_product_topic(ko.readme_example())
    with pytest.raises(ValueError):
        kao.evaluate(pt, pt.current[:3])                                     # ragged candidate
    one = ko.make_cluster("one", 3, 1, 1, 1, 1, [], []).topics[0]            # 1 partition, RF 1, 1 rack
    r = kao.solve([to_product_topic(one)], max_launches=2, restarts=4, iters_per_launch=16)[0]
    assert r.status == "OPTIMAL_PROVEN" and r.assignment.tolist() == one.current.tolist()
    full = ko.make_cluster("full", 3, 3, 1, 5, 3, [], []).topics[0]          # RF == B: every broker in every partition
    r = kao.solve([to_product_topic(full)], max_launches=4, restarts=4, iters_per_launch=64)[0]
    obj, viol = ko.verify(full, r.assignment)
    assert viol[0] == 0 and obj == r.objective == ko.solve_exact(full).objective


def test_device_bookkeeping_matches_verifier_on_edge_shapes(kao, ko):
    """Device restarts' tracked (objective, violation) equal the independent numpy verifier of their final
    state on structured edge shapes (single-broker racks, one rack, RF 1, RF = B - 1, tiny P)."""
    import itertools
    ots = []
    for B0, R, P, rf in itertools.product((2, 3, 4, 5, 7), (1, 2, 3, 4, 5), (1, 2, 5), (1, 2, 3)):
        if rf < B0 and R <= B0:
            ots.append(ko.make_cluster("e", B0, R, 1, P, rf, [], []).topics[0])
    with kao.Session([to_product_topic(t) for t in ots], seed=11, restarts=4, iters_per_launch=40) as s:
        s.step(2)
        assert s.stats()["drift"] == 0
        for ti, ot in enumerate(ots):
            st = s.restart_state(ti, ti % 4)
            obj, viol = ko.verify(ot, st["final"])
            assert (obj, int(viol[0])) == (st["obj"], st["V"]), (ti, ot.n_brokers, ot.n_racks, ot.n_partitions, ot.rf)
        for ot, r in zip(ots, s.best()):
            if r.status not in ("NO_FEASIBLE", "INFEASIBLE_PROVEN"):
                obj, viol = ko.verify(ot, r.assignment)
                assert viol[0] == 0 and obj == r.objective


def test_maximum_size_topic(kao, ko, kp):
    """Near the LDS limit: 4800 partitions x RF 3 = 14400 replicas on 40 brokers, one restart per workgroup."""
    ot = ko.make_cluster("max", 40, 4, 1, 4800, 3, [5], [(40, 1)]).topics[0]
    pt = to_product_topic(ot)
    cands = random_candidates(ot, 6, seed=3)
    obj, viol = kao.evaluate_batch(pt, cands)
    for i in range(len(cands)):
        o, v = kp.port_eval(ot, cands[i])
        assert (int(obj[i]), viol[i].tolist()) == (o, v.tolist())
    seed = 7
    with kao.Session([pt], seed=seed, restarts=4, iters_per_launch=48) as s:
        assert s.stats()["blocks_search"] == 4  # 1 wave per workgroup
        s.step(1)
        assert s.stats()["drift"] == 0
        dev = s.restart_state(0, 2)
        ref = kp.port_search(ot, seed ^ 0x9E3779B97F4A7C15, 2, 1, 48)
        assert dev["final"].tolist() == ref["final"].tolist()
        assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])


def test_unsupported_instances_are_rejected(kao, ko):
    from kafka_assignment_optimizer_amd import Topic
    huge = Topic(name="huge", broker_ids=np.arange(100), rack_of=np.zeros(100, dtype=np.uint8), n_racks=1,
                 n_partitions=1_400_000, rf=3, current=np.zeros((1_400_000, 1), dtype=np.uint16))
    with pytest.raises(kao.KaoError) as e:
        kao.derive_bounds(huge)
    assert e.value.code == -2  # 4.2 M replicas > 4,000,000
    dense = Topic(name="dense", broker_ids=np.arange(2), rack_of=np.zeros(2, dtype=np.uint8), n_racks=1,
                  n_partitions=70_000, rf=1, current=np.zeros((70_000, 1), dtype=np.uint16))
    with pytest.raises(kao.KaoError) as e:
        kao.derive_bounds(dense)
    assert e.value.code == -2  # 35,000 replicas per broker on average: 16-bit per-broker counters


def _oracle_topic(ko, pt):
    return ko.Topic(name=pt.name, broker_ids=pt.broker_ids, rack_of=pt.rack_of, n_racks=pt.n_racks, n_partitions=pt.n_partitions,
                    rf=pt.rf, current=pt.current, weights=pt.weights, bounds_override=dict(pt.bounds_override))


def test_topic_in_global_memory_replay_and_eval(kao, ko, kp, monkeypatch):
    """6000 partitions x 1000 brokers does not fit LDS with both assignments even with one restart per workgroup: the kernel keeps the
    assignment words in global memory (k_search<true>; KAO_CUR_GLOBAL=0 keeps round 5's k_search_curg, which would take this size, out
    of the way -- it has its own replay test).  Same spec: bit-exact against the scalar replay; K-eval on 12000 x 3 = 36000 replicas
    (> 32767: unpacked wave sums) bit-exact against the C evaluator."""
    monkeypatch.setenv("KAO_CUR_GLOBAL", "0")
    from kafka_assignment_optimizer_amd import synthetic
    pt = synthetic.make_cluster(1000, 20, 1, 6000, 3, [7, 77, 777], [(1000, 7), (1001, 17), (1002, 17)])[0]
    ot = _oracle_topic(ko, pt)
    seed = 2718
    with kao.Session([pt], seed=seed, restarts=4, iters_per_launch=40, team=1) as s:   # team=1: one wavefront per restart (k_search<true>)
        st = s.stats()
        assert st["lds_bytes_search"] < 40 * 1024  # only broker / rack tables in LDS
        s.step(2)
        assert s.stats()["drift"] == 0
        tseed = seed ^ 0x9E3779B97F4A7C15
        for rho in (0, 3):
            dev = s.restart_state(0, rho)
            ref = kp.port_search(ot, tseed, rho, 2, 40)
            assert dev["final"].tolist() == ref["final"].tolist()
            assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])
            obj, viol = kp.port_eval(ot, dev["final"])
            assert (obj, int(viol[0])) == (dev["obj"], dev["V"])
    big = synthetic.make_cluster(400, 8, 1, 12000, 3, [5], [(400, 5)])[0]
    ob = _oracle_topic(ko, big)
    cands = random_candidates(ob, 5, seed=4, p_mut=0.2, p_none=0.02)
    obj, viol = kao.evaluate_batch(big, cands)
    for i in range(len(cands)):
        o, v = kp.port_eval(ob, cands[i])
        assert (int(obj[i]), viol[i].tolist()) == (o, v.tolist())


@pytest.mark.parametrize("team", [8, 3])
def test_team_search_replay_bit_exact(kao, ko, kp, team):
    """Topics in global memory, round 4: a TEAM of wavefronts searches one restart (k_team) -- every wavefront proposes a move
    per iteration against the same state, proposals sharing no partition / broker / rack with a lower-numbered wavefront's
    are all applied.  Same deterministic spec as oracle/kao_port.c::ls_run(team): final state, best snapshot, V, objective and
    the number of applied moves bit for bit; with host-set prices and an elite launch too."""
    from kafka_assignment_optimizer_amd import synthetic
    pt = synthetic.drift(synthetic.make_cluster(1000, 20, 1, 6000, 3, [7, 77, 777], [(1000, 7), (1001, 17), (1002, 17)]), 0.2, 3)[0]
    ot = _oracle_topic(ko, pt)
    seed, iters = 1618 + team, 48
    tseed = seed ^ 0x9E3779B97F4A7C15
    with kao.Session([pt], seed=seed, restarts=3, iters_per_launch=iters, team=team) as s:
        assert s.stats()["lds_bytes_search"] < 40 * 1024
        s.step(3)
        assert s.stats()["drift"] == 0
        n_acc = []
        for rho in range(3):
            dev = s.restart_state(0, rho)
            ref = kp.port_search(ot, tseed, rho, 3, iters, team=team)
            assert np.array_equal(dev["final"], ref["final"]) and np.array_equal(dev["best"], ref["best"]), (team, rho)
            assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])
            obj, viol = kp.port_eval(ot, dev["final"])
            assert (obj, int(viol[0])) == (dev["obj"], dev["V"])
            n_acc.append(dev["n_accept"])
        single = kp.port_search(ot, tseed, 0, 3, iters)["n_accept"]
        assert n_acc[0] > 2 * single            # the team applies several moves per iteration
    # priced instantiation + an elite launch
    rng = np.random.default_rng(team)
    prices = ((rng.integers(-8, 9, ot.n_brokers) * 16384).astype(np.int32), (rng.integers(-4, 5, ot.n_brokers) * 16384 + rng.integers(-4800, 4800, ot.n_brokers)).astype(np.int32),
              (rng.integers(-2, 3, ot.n_racks) * 16384).astype(np.int32))
    with kao.Session([pt], seed=seed, restarts=3, iters_per_launch=iters, elite_period=2, team=team) as s:
        s.set_prices(0, *prices)
        s.step(2)
        r = s.best()[0]
        s.step(1)
        assert s.stats()["drift"] == 0
        for rho in range(3):
            run = kp.PortRun(ot, tseed, rho, team=team)
            run.launch(0, iters, prices=prices); run.launch(1, iters, prices=prices)
            run.launch(2, iters, prices=prices, elite=(r.assignment, r.objective, r.best_restart) if r.status != "NO_FEASIBLE" else None)
            ref = run.read()
            dev = s.restart_state(0, rho)
            assert np.array_equal(dev["final"], ref["final"]) and np.array_equal(dev["best"], ref["best"]), (team, rho)
            assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])


@pytest.mark.parametrize("shape", ["rf3", "rf6", "rf3_priced_team"])
def test_k_init_fills_the_holes_like_one_wavefront(kao, ko, kp, monkeypatch, shape):
    """Round 6: topics in global memory get their holes filled by K-init, one WORKGROUP per restart (the rounds of a hole dealt to its
    wavefronts, one LDS atomic min and one barrier per hole), instead of one wavefront inside k_search.  Same holes, same order, same
    winners: the restart states after the first launches are those of KAO_INIT_WAVES=0 (the old fill) for every team size, for the
    8-word instantiation (RF 6) and with host-set prices behind a re-initialisation; and they are the scalar replay's (kao_port.c)."""
    from kafka_assignment_optimizer_amd import synthetic
    rf = 6 if shape == "rf6" else 3
    gone = [3, 33, 133, 233, 333, 433, 533, 633, 733, 833]
    pt = synthetic.make_cluster(1000, 20, 1, 6000, rf, gone, [(1000 + i, (7 * i) % 20) for i in range(8)])[0]
    ot = _oracle_topic(ko, pt)
    seed, iters, team = 4242, 24, (4 if shape.endswith("team") else 1)
    rng = np.random.default_rng(6)
    prices = ((rng.integers(-8, 9, ot.n_brokers) * 16384).astype(np.int32), (rng.integers(-4, 5, ot.n_brokers) * 16384).astype(np.int32),
              (rng.integers(-2, 3, ot.n_racks) * 16384).astype(np.int32)) if "priced" in shape else None
    monkeypatch.setenv("KAO_CUR_GLOBAL", "0")
    sigs = {}
    for waves in ("0", "1", "2", "8", "16", None):
        if waves is None: monkeypatch.delenv("KAO_INIT_WAVES", raising=False)
        else: monkeypatch.setenv("KAO_INIT_WAVES", waves)
        with kao.Session([pt], seed=seed, restarts=3, iters_per_launch=iters, team=team) as s:
            assert s.stats()["lds_bytes_search"] < 48 * 1024            # the global-memory path
            if prices is not None: s.set_prices(0, *prices)
            s.step(2)
            st = [s.restart_state(0, rho) for rho in range(3)]
        sigs[waves] = [(d["final"].tobytes(), d["best"].tobytes(), d["best_obj"], d["V"], d["obj"], d["n_accept"]) for d in st]
        if waves is None:
            tseed = seed ^ 0x9E3779B97F4A7C15
            for rho in (0, 2):
                run = kp.PortRun(ot, tseed, rho, team=team)
                run.launch(0, iters, prices=prices); run.launch(1, iters, prices=prices)
                ref = run.read()
                assert np.array_equal(st[rho]["final"], ref["final"]) and (st[rho]["V"], st[rho]["obj"], st[rho]["n_accept"]) == (ref["V"], ref["obj"], ref["n_accept"])
    assert all(v == sigs["0"] for v in sigs.values()), [k for k, v in sigs.items() if v != sigs["0"]]


def test_config5_as_one_topic(kao, ko, kp):
    """BASELINE config 5 taken literally as ONE topic: 1000 brokers, 20 racks, 100,000 partitions, RF 3, 50 brokers
    replaced, per-broker cap ceil(avg)+1 (north_star: time-to-optimal <= 1 s).  The topic (1.6 MB of assignment words
    per restart) runs on the global-memory path; measured: proven optimal (760000 = bound) in 0.105 s -- almost all of
    it the best-insertion fill of 15,000 holes (as 1000 topics x 100 partitions the same cluster takes 3.9 ms)."""
    from kafka_assignment_optimizer_amd import synthetic
    rng = synthetic.SplitMix64(synthetic.CONFIG_SEED + 5)
    rm = rng.sample(list(range(1000)), 50)
    add = [(1000 + i, b % 20) for i, b in enumerate(rm)]  # each new broker joins the rack of a removed one: with uneven
    # racks the single-topic rack band (exactly 15,000 per rack) would be infeasible
    pt = synthetic.make_cluster(1000, 20, 1, 100_000, 3, rm, add, bounds_override={"rep_hi": 301})[0]
    ot = _oracle_topic(ko, pt)
    kao.solve([pt], seed=1, restarts=64, iters_per_launch=16, max_launches=1)  # warm allocation of the big arenas
    import time
    t0 = time.perf_counter()
    r = kao.solve([pt], seed=5, restarts=64, iters_per_launch=128, stop_at_bound=1, time_limit_s=1.0)[0]
    dt = time.perf_counter() - t0
    tm = kao.last_solve_timing()
    print(f"cfg5 as one topic: {r.status} objective {r.objective} bound {r.upper_bound} launches {tm['launches']} "
          f"time_to_best {tm['time_to_best']:.3f}s total {dt:.3f}s")
    obj, viol = kp.port_eval(ot, r.assignment)
    assert viol[0] == 0 and obj == r.objective <= r.upper_bound          # feasible under the independent evaluator
    assert r.status == "OPTIMAL_PROVEN" and r.objective == r.upper_bound == 760000
    # north_star: time-to-optimal <= 1 s.  Asserted by counts, not by the clock: the proof comes with the init of the FIRST
    # launch -- no search iteration, no K-bound, no LP -- and that launch is 0.1 s on an MI355X (printed above; bench.py reports it)
    assert tm["launches"] == 1 and tm["lp_solves"] == 0


def test_drifted_north_star_topic_gets_a_dual_certificate(kao, ko, kp):
    """Round 4: the north-star size after a 20 % drift (1000 brokers x 100,000 partitions: nothing the init could simply keep).
    K-bound's limit on P * RF moved from 2^17 to 2^20, so the topic gets a Lagrangian certificate instead of the closed-form
    bound (786,857; first GPU run: 782,651 after 3 s with the incumbent at 782,074).  No exact solver can check the optimum at
    this size: what is asserted is the sandwich incumbent <= certificate < closed-form bound and that the gap is small."""
    from kafka_assignment_optimizer_amd import synthetic
    pt = synthetic.north_star_topic("drift100k")
    ot = _oracle_topic(ko, pt)
    closed = kao.upper_bound(pt)
    kao.solve([pt], seed=1, max_launches=1)   # warm allocation of the big arenas
    r = kao.solve([pt], seed=3, stop_at_bound=1, time_limit_s=3.0)[0]
    tm = kao.last_solve_timing()
    print(f"drifted 1000 x 100k: {r.status} objective {r.objective} certificate {r.upper_bound} (closed form {closed}) "
          f"launches {tm['launches']} K-bound iterations {tm['bound_iters']}")
    obj, viol = kp.port_eval(ot, r.assignment)
    assert viol[0] == 0 and obj == r.objective <= r.upper_bound
    assert tm["bound_iters"] > 0 and r.upper_bound < closed - 2000
    # round 4 (KAO-CX after every launch in bulk rounds): gaps 155-183 of 782,6xx after 3 s on four runs, 201-279 after 1 s.
    # Round 5: with room for it (limit >= 1.8 s) the perturbed LP runs straight after the first feasible incumbent; its row duals give
    # the certificate 782,512 and its rounded iterate an assignment of exactly that value: PROVEN in 1.65 s (GPU call 34)
    assert (r.status, r.objective, r.upper_bound) == ("OPTIMAL_PROVEN", 782512, 782512), (r.status, r.objective, r.upper_bound)
    # Round 6: an interior-point iteration costs 4.4 ms instead of 9.3 and the schedule asks a count-keyed estimate of the LP's time, not a
    # fixed limit: the LP runs before any K-search launch under the north-star's own budget too.  Counts, not the clock: ONE
    # solve of at most 200 iterations, its rounded iterate adopted, no K-search launch needed before it.
    lp = kao.last_solve_lp()
    assert lp["adopted"] == 1 and lp["solves"] == 1 and lp["iterations"] <= 200 and tm["launches"] <= 2, (lp, tm["launches"])
    r1 = kao.solve([pt], seed=3, stop_at_bound=1, time_limit_s=1.0)[0]          # north_star: <= 1 s time-to-optimal
    tm1, lp1 = kao.last_solve_timing(), kao.last_solve_lp()
    print(f"   under time_limit_s = 1.0: {r1.status} objective {r1.objective} certificate {r1.upper_bound} in {tm1['results_read_back']:.3f} s, {int(lp1['iterations'])} LP iterations")
    obj1, viol1 = kp.port_eval(ot, r1.assignment)
    assert viol1[0] == 0 and obj1 == r1.objective <= r1.upper_bound
    assert (r1.status, r1.objective, r1.upper_bound) == ("OPTIMAL_PROVEN", 782512, 782512), (r1.status, r1.objective, r1.upper_bound)


def test_large_topic_fewer_waves_per_workgroup(kao, ko, kp):
    """A 3000-partition topic (9000 replicas, 1000 brokers) does not fit LDS with four restarts per workgroup;
    the same kernel runs it with two, and the scalar replay still matches bit for bit."""
    ot = ko.make_cluster("big3000", 1000, 20, 1, 3000, 3, [7, 77, 777], [(1000, 7), (1001, 17), (1002, 17)]).topics[0]
    seed = 31337
    with kao.Session([to_product_topic(ot)], seed=seed, restarts=6, iters_per_launch=400) as s:
        st = s.stats()
        assert st["lds_bytes_search"] <= 160 * 1024 and st["blocks_search"] == 3  # 2 waves per workgroup instead of 4
        s.step(2)
        assert s.stats()["drift"] == 0
        tseed = seed ^ (0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
        for rho in (0, 5):
            dev = s.restart_state(0, rho)
            ref = kp.port_search(ot, tseed, rho, 2, 400)
            assert dev["final"].tolist() == ref["final"].tolist()
            assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])
        r = s.best()[0]
    if r.status not in ("NO_FEASIBLE", "INFEASIBLE_PROVEN"):
        obj, viol = kp.port_eval(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective


# ------------------------------------------------------------------------------- optimum parity
def test_kat1_end_to_end(kao, ko):
    """README.md:52-63 in -> README.md:88 out ([8,19] -> [8,1], nothing else moves)."""
    import kafka_assignment_optimizer_amd as k
    racks = {b: ("a" if b % 2 == 0 else "b") for b in range(20)}
    topics = k.topics_from_json(ko.README_CURRENT, list(range(19)), racks)
    res = k.solve(topics, seed=1, time_limit_s=5.0, stop_at_bound=1, restarts=64, iters_per_launch=128)
    r = res[0]
    assert r.status == "OPTIMAL_PROVEN" and r.objective == 58 and r.upper_bound == 58
    canon = k.canonicalize(topics[0], r.assignment)
    g = load_golden("kat1.json")
    assert canon.tolist() == g["expected_assignment"]
    assert k.assignment_to_json(topics, [canon]) == g["expected_json"]
    assert ko.count_moves(ko.readme_example(), canon) == (1, 0)


def test_alternative_weight_scheme(kao, ko):
    """the README fixes only the weight multiset {1,2,2,4}; scheme C (LL=4, LF=2, FL=2, FF=1)
    gives 49 on KAT-1 and must be honoured end to end (objective, evaluation, bound)."""
    ot = ko.readme_example()
    ot.weights = ((4, 2), (2, 1))
    pt = to_product_topic(ot)
    r = kao.solve([pt], seed=2, restarts=64, iters_per_launch=128, stop_at_bound=1, time_limit_s=5)[0]
    assert (r.status, r.objective, r.upper_bound) == ("OPTIMAL_PROVEN", 49, 49)
    obj, viol = ko.verify(ot, r.assignment)
    assert viol[0] == 0 and obj == 49 and ko.count_moves(ot, r.assignment) == (1, 0)
    for s in range(20, 40):  # random instances under scheme C: exact objective parity
        o2 = ko.random_case(s, max_b=12, max_p=8)
        if o2.rf > 8 or o2.rf_cur > 8:
            continue
        o2.weights = ((4, 2), (2, 1))
        ex = ko.solve_exact(o2, 30)
        r = kao.solve([to_product_topic(o2)], seed=s, restarts=32, iters_per_launch=256, max_launches=6, time_limit_s=10)[0]
        if ex.status == "infeasible":
            assert r.status in ("NO_FEASIBLE", "INFEASIBLE_PROVEN")
        else:
            assert r.objective == ex.objective <= r.upper_bound, s


def test_canonicalize_matches_oracle(kao, ko):
    """kao_canonicalize (k_canon on the device) == the oracle's canonicalize on solved instances: golden random cases,
    BASELINE config topics, and a larger topic (many new placements, several passes)."""
    ots = [ko.topic_from_dict(c["topic"]) for c in load_golden("random_small.json")["cases"] if c["status"] == "optimal"]
    ots += [ko.gen_config(n, n_topics=2).topics[1] for n in (2, 3, 4)]
    ots.append(ko.make_cluster("rf2to3", 40, 4, 1, 90, 2, [5, 6], [(40, 1), (41, 1), (42, 2)], new_rf=3).topics[0])
    pts = [to_product_topic(t) for t in ots]
    res = kao.solve(pts, seed=17, iters_per_launch=256, max_launches=6, stop_at_bound=1, time_limit_s=20)
    n = 0
    for ot, pt, r in zip(ots, pts, res):
        if r.status in ("NO_FEASIBLE", "INFEASIBLE_PROVEN"):
            continue
        got = kao.canonicalize(pt, r.assignment)
        want = ko.canonicalize(ot, r.assignment)
        assert got.tolist() == want.tolist(), ot.name
        obj, viol = ko.verify(ot, got)
        assert viol[0] == 0 and obj == r.objective
        again = kao.canonicalize(pt, got)
        assert again.tolist() == got.tolist()  # idempotent
        n += 1
    assert n >= 30
    bad = res[0].assignment.copy()
    bad[0, 0] = bad[0, -1] if ots[0].rf > 1 else 0xFFFF  # infeasible input is returned untouched
    if ko.verify(ots[0], bad)[1][0] != 0:
        assert kao.canonicalize(pts[0], bad).tolist() == bad.tolist()


def test_golden_optima_random_small(kao, ko):
    cases = [c for c in load_golden("random_small.json")["cases"]]
    ots = [ko.topic_from_dict(c["topic"]) for c in cases]
    pts = [to_product_topic(t) for t in ots]
    res = kao.solve(pts, seed=5, restarts=32, iters_per_launch=512, max_launches=8, time_limit_s=30.0)
    n_unique = 0
    for c, ot, pt, r in zip(cases, ots, pts, res):
        if c["status"] == "infeasible":
            assert r.status == "INFEASIBLE_PROVEN", c["seed"]  # every golden infeasible case is caught by counting
            continue
        assert r.objective == c["objective"], (c["seed"], r.objective, c["objective"])
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective
        assert r.violations.tolist() == [0] * 8 and r.objective <= r.upper_bound
        if c.get("unique"):
            # unique optimum: the 0/1 vector is determined -> leader and follower SET bit-exact;
            # follower order is fixed by the canonical form
            canon = kao.canonicalize(pt, r.assignment)
            want = ko.canonicalize(ot, np.array(c["assignment"]))
            assert canon.tolist() == want.tolist(), c["seed"]
            n_unique += 1
    assert n_unique >= 5


def test_golden_optima_random_medium(kao, ko):
    """100 medium random instances (up to 40 brokers x 40 partitions, RF changes, adds and removals): the device
    objective equals the HiGHS optimum, infeasible instances are reported as such, the bound never undercuts the
    optimum, and unique optima are reproduced bit for bit (after the canonical follower order)."""
    cases = load_golden("random_medium.json")["cases"]
    ots = [ko.topic_from_dict(c["topic"]) for c in cases]
    pts = [to_product_topic(t) for t in ots]
    res = kao.solve(pts, seed=23, restarts=64, iters_per_launch=512, max_launches=10, time_limit_s=60.0)
    n_opt = n_unique = n_proven = 0
    for c, ot, pt, r in zip(cases, ots, pts, res):
        if c["status"] == "infeasible":
            assert r.status == "INFEASIBLE_PROVEN", c["seed"]  # every golden infeasible case is caught by counting
            continue
        assert r.objective == c["objective"], (c["seed"], r.objective, c["objective"])
        assert c["objective"] <= r.upper_bound
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective
        n_opt += 1
        n_proven += r.status == "OPTIMAL_PROVEN"
        if c.get("unique"):
            assert kao.canonicalize(pt, r.assignment).tolist() == ko.canonicalize(ot, np.array(c["assignment"])).tolist(), c["seed"]
            n_unique += 1
    assert n_opt >= 70 and n_unique >= 5 and n_proven == n_opt, (n_opt, n_proven)   # 77 of 77 proven since round 4 (tools/tol_probe.py); an equality since round 5


@pytest.mark.parametrize("name", ["cfg2.json", "cfg3.json", "cfg4.json"])
def test_golden_optima_configs(kao, ko, name):
    g = load_golden(name)
    ots = [ko.topic_from_dict(e["topic"]) for e in g["topics"]]
    res = kao.solve([to_product_topic(t) for t in ots], seed=11, iters_per_launch=512, max_launches=12,
                    time_limit_s=30.0)
    for e, ot, r in zip(g["topics"], ots, res):
        assert r.objective == e["objective"], (name, r.objective, e["objective"], r.status)
        assert r.status == "OPTIMAL_PROVEN" and r.upper_bound == e["objective"]  # the bound certificate is tight here
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == e["objective"]
        assert ko.count_moves(ot, r.assignment)[0] == e["moves"][0]


# ------------------------------------------------------------------------------- full-size properties
@pytest.mark.parametrize("cfg", [3, 4])
def test_full_config_properties(kao, ko, kp, cfg):
    case = ko.gen_config(cfg)
    ots = case.topics
    pts = [to_product_topic(t) for t in ots]
    with kao.Session(pts, seed=2024, iters_per_launch=512) as s:
        s.step(6)
        res = s.best()
        st = s.stats()
    assert st["drift"] == 0 and st["launches"] == 6
    full = load_golden(f"cfg{cfg}_full.json")["topics"]   # HiGHS optimum of EVERY topic (tests/golden/make_golden_full.py)
    assert len(full) == len(ots)
    for i, e in enumerate(load_golden(f"cfg{cfg}.json")["topics"]):
        assert e["objective"] == full[i]["objective"]     # the two golden files agree
    again = []
    for i, (ot, r) in enumerate(zip(ots, res)):
        obj, viol = kp.port_eval(ot, r.assignment)
        assert viol[0] == 0, (cfg, i, viol)           # feasible under the independent verifier
        assert obj == r.objective <= r.upper_bound    # reported objective is the true objective
        assert r.objective == full[i]["objective"], (cfg, i, r.objective, full[i]["objective"])   # == exact optimum, all topics
        assert r.upper_bound >= full[i]["objective"]  # the certificate never undercuts the exact optimum
        assert r.status == "OPTIMAL_PROVEN" and r.upper_bound == full[i]["objective"]
        assert ko.count_moves(ot, r.assignment)[0] == full[i]["moves"][0]   # as few replica moves as the exact solver
        if i < 8:  # idempotence: the solution, fed back as the current assignment, is a fixed point
            t2 = to_product_topic(ot)
            t2.current = r.assignment.copy()
            again.append((t2, r))
    res2 = kao.solve([t for t, _ in again], seed=3, iters_per_launch=256, max_launches=4, stop_at_bound=1, time_limit_s=20)
    for (t2, r), r2 in zip(again, res2):
        assert r2.status == "OPTIMAL_PROVEN"
        moved = sum(len(set(a) - set(b)) for a, b in zip(r2.assignment.tolist(), r.assignment.tolist()))
        assert moved == 0 and r2.assignment[:, 0].tolist() == r.assignment[:, 0].tolist()


def test_config5_sample_and_caps(kao, ko, kp):
    """cfg5 (1000 brokers, 20 racks, per-broker load caps): 64 of the 1000 topics."""
    ots = ko.gen_config(5, n_topics=64).topics
    assert ots[0].bounds()["rep_hi"] == 2  # cap = ceil(avg)+1
    res = kao.solve([to_product_topic(t) for t in ots], seed=9, iters_per_launch=512, max_launches=6, time_limit_s=30)
    for ot, r in zip(ots, res):
        obj, viol = kp.port_eval(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective
    full = load_golden("cfg5_full.json")["topics"]        # HiGHS optimum of each of the 64 topics
    assert len(full) == len(ots)
    for i, (ot, r) in enumerate(zip(ots, res)):
        assert r.objective == full[i]["objective"], (i, r.objective, full[i]["objective"])
        assert r.upper_bound >= full[i]["objective"]
        assert r.status == "OPTIMAL_PROVEN"


def test_determinism(kao, ko):
    pts = [to_product_topic(t) for t in ko.gen_config(4, n_topics=4).topics]
    outs = []
    for _ in range(2):
        with kao.Session(pts, seed=99, restarts=16, iters_per_launch=200) as s:
            s.step(3)
            outs.append(([r.assignment.tolist() for r in s.best()], s.best_keys().tolist()))
    assert outs[0] == outs[1]


def _drift_topic(B, R, P, dseed=1):
    """The drifted single topics of tools/drift_scale.py / tests/golden/drift_scale.json (product-side generator)."""
    from kafka_assignment_optimizer_amd import synthetic as sy
    return sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, dseed)[0]


def test_solve_is_deterministic(kao, ko):
    """kao_solve under its default (deterministic) schedule: K-bound targets and launch lengths, price adoption, KAO-CX calls and
    elite launches are keyed to launch / iteration / round counts, never to the clock (the answer on large topics
    depended on launch timing).  Same seed => same assignment, certificate, launch count, K-bound iterations and KAO-CX calls --
    on a topic that ends in a proof and on one stopped by max_launches with a gap left."""
    for (B, R, P, ml) in ((100, 5, 1000, 0), (300, 6, 2000, 150)):
        t = _drift_topic(B, R, P)
        outs = []
        for _ in range(3):
            r = kao.solve([t], seed=11, time_limit_s=60.0, max_launches=ml)[0]
            tm = kao.last_solve_timing()
            outs.append((r.status, r.objective, r.upper_bound, r.assignment.tolist(), tm["launches"], tm["bound_launches"], tm["bound_iters"],
                         tm["cx_calls"], tm["cx_gains"], tm["search_iters"], tm["lp_solves"], tm["lp_iters"]))
        assert outs[0] == outs[1] == outs[2], [(o[:3], o[4:]) for o in outs]
        assert outs[0][6] > 0 and (ml == 0 or outs[0][4] <= ml)
        if ml:   # round 5: the 300 x 2000 run is long enough for KAO-LP (interior-point iterations beside the launches): counted, and the same every time
            assert outs[0][10] == 1 and outs[0][11] > 5


@pytest.mark.parametrize("B,R,P", [(100, 5, 1000), (200, 5, 2000), (300, 6, 2000)])
def test_drift_scale_optima_are_reached_and_proven(kao, ko, B, R, P):
    """tests/golden/drift_scale.json: drifted single topics of 1,000-2,000 partitions whose exact optimum HiGHS established
    (MILP optimum where branch and bound finished, else the value of the LP relaxation, which the certificate meets): kao_solve
    returns THAT objective, proven, for three seeds (the driver's bench showed 14824 against 14826) -- by the
    first population or, when that one converges a unit short, by a later generation."""
    row = [r for r in load_golden("drift_scale.json")["rows"] if (r["B"], r["R"], r["P"]) == (B, R, P)][0]
    opt = row.get("milp_objective") or int(row["lp_value"])
    assert float(opt) == row["lp_value"]
    t = _drift_topic(B, R, P)
    ot = ko.Topic(name=t.name, broker_ids=np.array(t.broker_ids), rack_of=np.array(t.rack_of), n_racks=t.n_racks,
                  n_partitions=t.n_partitions, rf=t.rf, current=np.array(t.current), weights=t.weights)
    for seed in (1, 2, 3):
        r = kao.solve([t], seed=seed, time_limit_s=20.0)[0]
        assert (r.status, r.objective, r.upper_bound) == ("OPTIMAL_PROVEN", opt, opt), (seed, r.status, r.objective, r.upper_bound)
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == opt


def test_slack_band_certificate_meets_the_lp_value(kao, ko):
    """Round 3: a topic whose replicas do not divide evenly (6,600 on 270 brokers: bands 24..25 -- every drifted topic of the
    family above has P * RF a multiple of B).  The dual function is piecewise linear along the common shift of a family of
    multipliers and subgradient steps crawl there: the certificate stood at 16471 after 3 s against an LP value of 16459
    (HiGHS interior point, tests/golden/drift_scale.json).  With the exact line search along the shifts at every launch
    (k_bound_center) it meets the LP value; the incumbent stays a feasible assignment below it."""
    row = [r for r in load_golden("drift_scale.json")["rows"] if (r["B"], r["R"], r["P"]) == (270, 6, 2200)][0]
    t = _drift_topic(270, 6, 2200)
    ot = ko.Topic(name=t.name, broker_ids=np.array(t.broker_ids), rack_of=np.array(t.rack_of), n_racks=t.n_racks,
                  n_partitions=t.n_partitions, rf=t.rf, current=np.array(t.current), weights=t.weights)
    lp = int(round(row["lp_value"]))
    # round 5 (KAO-LP: the certificate comes from the LP's own duals after 13 interior-point iterations, K-bound then leaves the
    # topic to the search): seeds 3 / 4 proven after 241 / 433 launches (2.2 / 3.7 s on the GPU box).  Round 4: 16457 on seed 3 after 4 s
    r = kao.solve([t], seed=3, time_limit_s=16.0, stop_at_bound=1)[0]
    obj, viol = ko.verify(ot, r.assignment)
    assert viol[0] == 0 and obj == r.objective
    assert (r.status, r.objective, r.upper_bound) == ("OPTIMAL_PROVEN", lp, lp), (r.status, r.objective, r.upper_bound)
    for seed in (1, 2):
        r = kao.solve([t], seed=seed, time_limit_s=16.0, stop_at_bound=1)[0]   # (proven inside 8 s on the GPU box; the schedule is count-keyed, the limit only has to be generous)
        assert (r.status, r.objective, r.upper_bound) == ("OPTIMAL_PROVEN", lp, lp), (seed, r.status, r.objective, r.upper_bound)


@pytest.mark.parametrize("B,R,P", [(350, 7, 2500), (450, 9, 3500)])
def test_more_slack_band_topics(kao, ko, B, R, P):
    """The other slack-band topics VERDICT r03 named (incumbents 8-21 units below the LP value then).  350 x 2500 (LP 18751) and
    450 x 3500 (LP 26330): PROVEN optimal for three seeds.  Round 4 reached 26330 on the second one but K-bound's certificate
    stalled at 26336-7 (the subgradient method on the kinks of three slack-band families) and the status stayed TIME_LIMIT; round 5's
    KAO-LP takes the multipliers from the LP itself (16 interior-point iterations beside the first launches): certificate 26330."""
    row = [r for r in load_golden("drift_scale.json")["rows"] if (r["B"], r["R"], r["P"]) == (B, R, P)][0]
    lp = int(round(row["lp_value"]))
    t = _drift_topic(B, R, P)
    for seed in (1, 2, 3):
        r = kao.solve([t], seed=seed, time_limit_s=16.0, stop_at_bound=1)[0]   # (proven inside 2 s on the GPU box; the schedule is count-keyed, the limit only has to be generous)
        assert (r.status, r.objective, r.upper_bound) == ("OPTIMAL_PROVEN", lp, lp), (seed, r.status, r.objective, r.upper_bound)


def test_further_kao_cx_starts(kao, ko, monkeypatch):
    """Round 3: besides the elite, KAO-CX descends from the best snapshots of other restarts (which basin it ends in is decided by
    where it starts).  The drifted 300 x 2000 topic (MILP optimum 14826; round 2's bench returned 14824) is proven with them,
    they are counted in the solve's timing record, KAO_DET_CX_STARTS=0 switches them off, and on the second drift seed of the
    same shape (MILP optimum 14801, drift_scale.json rows_other_seeds) certificate and incumbent stay within two units."""
    t = _drift_topic(300, 6, 2000)
    monkeypatch.setenv("KAO_LP_ROUND", "0")     # the search engines alone (with KAO-LP's rounded iterate the topic is proven before KAO-CX is called)
    r = kao.solve([t], seed=3, time_limit_s=20.0)[0]
    tm = kao.last_solve_timing()
    assert (r.status, r.objective, r.upper_bound) == ("OPTIMAL_PROVEN", 14826, 14826)
    assert 1 <= tm["cx_further_starts"] < tm["cx_calls"]
    monkeypatch.delenv("KAO_LP_ROUND")
    other = load_golden("drift_scale.json")["rows_other_seeds"][0]
    from kafka_assignment_optimizer_amd import synthetic as sy
    t2 = sy.drift(sy.make_cluster(other["B"], other["R"], 1, other["P"], 3, [], []), 0.2, other["seed"])[0]
    # round 5, second half: the rounded iterate of the perturbed LP IS the optimum -- every solver seed proves 14801 (rounds 3-5 with the
    # search engines alone: three to four seeds of five, the others a unit short after 8 s: VERDICT r04's open item)
    for seed in (1, 2, 3, 4, 5):
        r2 = kao.solve([t2], seed=seed, time_limit_s=8.0, stop_at_bound=1)[0]
        assert (r2.status, r2.objective, r2.upper_bound) == ("OPTIMAL_PROVEN", other["milp_objective"], other["milp_objective"]), (seed, r2.status, r2.objective)
    monkeypatch.setenv("KAO_LP_ROUND", "0")
    monkeypatch.setenv("KAO_DET_CX_STARTS", "0")
    kao.solve([t], seed=3, time_limit_s=20.0, max_launches=200)
    assert kao.last_solve_timing()["cx_further_starts"] == 0


@pytest.mark.parametrize("B,R,P", [(400, 8, 3000), (250, 5, 4000)])
def test_drift_scale_certificates_meet_the_lp_value(kao, ko, B, R, P):
    """The larger rows of drift_scale.json (only the LP relaxation finished on the CPU: 2,438 s / hours): the device proves the
    optimum -- incumbent == certificate == floor(LP value) -- for three seeds (round 4; the limit only decides how many launches
    of the count-keyed schedule fit: all six solves ended inside 6 s on the GPU box)."""
    row = [r for r in load_golden("drift_scale.json")["rows"] if (r["B"], r["R"], r["P"]) == (B, R, P)][0]
    t = _drift_topic(B, R, P)
    for seed in (1, 2, 3):
        r = kao.solve([t], seed=seed, time_limit_s=16.0, stop_at_bound=1)[0]
        # round 4: proven for all three seeds inside 6 s (tools/tol_probe.py, GPU call 14); round 3 accepted a certificate one unit
        # above floor(LP) and incumbents up to 12 units below it
        assert (r.status, r.objective, r.upper_bound) == ("OPTIMAL_PROVEN", int(row["lp_value"]), int(row["lp_value"])), (seed, r.status, r.objective, r.upper_bound)


# ------------------------------------------------------------------------------- K-bound (Lagrangian dual certificate)
def _wide_cases(ko, status="optimal"):
    return [(c, ko.random_case_wide(c["seed"])) for c in load_golden("random_wide.json")["cases"] if c["status"] == status]


def _bound_chunk(monkeypatch, chunk):
    """KAO_BOUND_CHUNK / KAO_BOUND_MULTI (test hooks).  None = the library's own choice (one persistent workgroup per topic below
    1,024 partitions, k_bound_multi -- persistent, slices of 256 partitions, one barrier per iteration -- beyond), "0" = always
    k_bound's single workgroup, "64" / "192" = always sliced that small on k_bound_multi (several workgroups even for the small
    families), "step64" / "step" = the round-2 driver k_bound_step (one kernel launch per iteration), slices of 64 / its own choice."""
    monkeypatch.delenv("KAO_BOUND_CHUNK", raising=False)
    monkeypatch.delenv("KAO_BOUND_MULTI", raising=False)
    if chunk is None:
        return
    if chunk.startswith("step"):
        monkeypatch.setenv("KAO_BOUND_MULTI", "0")
        chunk = chunk[4:]
    if chunk:
        monkeypatch.setenv("KAO_BOUND_CHUNK", chunk)


@pytest.mark.parametrize("chunk", [None, "64", "192", "step64"])
def test_dual_bound_replay_bit_exact(kao, ko, kp, monkeypatch, chunk):
    """K-bound vs its scalar replay (oracle/kao_port.c::kao_port_dual_bound): identical multipliers, best dual value,
    iteration count and stop flags -- one launch, and several launches that continue from the state in HBM.  Both drivers:
    the persistent workgroup and the sliced one-iteration-per-launch kernel."""
    _bound_chunk(monkeypatch, chunk)
    picked = [(c, t) for c, t in _wide_cases(ko) if c["upper_bound"] != c["objective"]][:40]
    picked += [(e, ko.topic_from_dict(e["topic"])) for n in ("cfg2.json", "cfg3.json", "cfg4.json") for e in load_golden(n)["topics"][:1]]
    assert len(picked) >= 40
    for i, (c, ot) in enumerate(picked):
        target = c["objective"] - (i % 3 == 2) * 2      # every third case aims at a suboptimal incumbent
        iters, launches = (37, 3) if i % 2 else (120, 1)
        got = kao.dual_bound(to_product_topic(ot), target, iters=iters, launches=launches)
        st = kp.DualState(ot)
        for _ in range(launches):
            st = kp.port_dual_bound(ot, target, iters, st)
            if st.flags & 7:
                break
        tag = (i, c.get("seed"), ot.n_brokers, ot.n_partitions)
        assert (got["iters"], got["flags"]) == (st.iters, st.flags), tag
        assert got["best_dual"] == st.best_L, tag
        assert got["a"].tolist() == st.a.tolist() and got["l"].tolist() == st.l.tolist() and got["g"].tolist() == st.g[:ot.n_racks].tolist(), tag
        assert got["bound"] == st.bound >= c["objective"], tag


@pytest.mark.parametrize("chunk", [None, "64", "step64"])
def test_dual_bound_replay_high_rf_and_broker_weights(kao, ko, kp, monkeypatch, chunk):
    """Round 3: K-bound beyond the README's range -- RF 5..8 (k_bound<8>: 8 replica slots per lane, 8 + 9 candidates per rack)
    and broker weights (plain objective coefficients inside the priced values) -- against the scalar replay, which solves every
    partition subproblem by brute force over all brokers: identical multipliers, dual value, iteration count, flags; and the
    certificate never undercuts the HiGHS optimum."""
    _bound_chunk(monkeypatch, chunk)
    rng = np.random.default_rng(77)
    picked = [(c, ko.random_case_rf(c["seed"])) for c in load_golden("random_rf.json")["cases"] if c["status"] == "optimal"][:36]
    weighted = []
    for c, t in [(c, t) for c, t in _wide_cases(ko) if c["upper_bound"] != c["objective"]][:16] + picked[:8]:
        t = ko.random_case_wide(c["seed"]) if "shape" in c else ko.random_case_rf(c["seed"])
        t.broker_w = rng.integers(0, 6, t.n_brokers).astype(np.int32)
        t.broker_wl = rng.integers(0, 4, t.n_brokers).astype(np.int32) if len(weighted) % 3 else None
        weighted.append((None, t))
    n_rf = n_w = 0
    for i, (c, ot) in enumerate(picked + weighted):
        opt = c["objective"] if c is not None else ko.upper_bound_simple(ot) - 3
        target = max(0, opt - (i % 3 == 2) * 2)
        iters, launches = (37, 3) if i % 2 else (120, 1)
        got = kao.dual_bound(to_product_topic(ot), target, iters=iters, launches=launches)
        st = kp.DualState(ot)
        for _ in range(launches):
            st = kp.port_dual_bound(ot, target, iters, st)
            if st.flags & 7:
                break
        tag = (i, ot.n_brokers, ot.n_partitions, ot.rf, ot.rf_cur)
        assert (got["iters"], got["flags"]) == (st.iters, st.flags), tag
        assert got["best_dual"] == st.best_L, tag
        assert got["a"].tolist() == st.a.tolist() and got["l"].tolist() == st.l.tolist() and got["g"].tolist() == st.g[:ot.n_racks].tolist(), tag
        if c is not None:
            assert got["bound"] == st.bound >= c["objective"], tag
            n_rf += ot.rf > 4 or ot.rf_cur > 4
        else:
            n_w += 1
    assert n_rf >= 20 and n_w >= 20


@pytest.mark.parametrize("chunk", [None, "0", "64", "step", "step64"])
def test_dual_bound_replay_large_shapes(kao, ko, kp, monkeypatch, chunk):
    """(chunk None: the 20,000-partition shape runs on k_bound_multi sliced over 79 workgroups, the sub-1,024-partition shapes on
    k_bound's one workgroup; "0": all on one workgroup; "64": all sliced on k_bound_multi; "step" / "step64": k_bound_step, one
    launch per iteration.)  K-bound paths the small families do not reach: the current assignment read from global memory (8 B per
    partition no longer fits LDS), hundreds of brokers in few racks (long per-rack scans), many racks (rack ranking
    over several 64-lane rounds), RF 4 with an RF change; multipliers and dual value still equal the replay's."""
    shapes = [  # (brokers, racks, partitions, rf, removed, added, new_rf, target offset)
        (50, 5, 20000, 3, [3, 7], [(50, 1)], None, 40),
        (900, 3, 300, 3, list(range(0, 90, 7)), [(900 + i, i % 3) for i in range(5)], None, 25),
        (400, 200, 150, 2, [5, 6, 7], [(400, 3)], None, 1),
        (64, 4, 500, 4, [1], [(64, 2), (65, 3)], 3, 3),
        (130, 1, 257, 1, [0, 129], [], 2, 5),
        (60, 4, 50000, 3, [3], [(60, 1)], None, 60),     # round 4: 150,000 replica slots -- beyond the 2^17 of rounds 1-3 (kao_model.cpp dual_supported)
    ]
    for i, (B, R, P, rf, rm, add, new_rf, off) in enumerate(shapes):
        ot = ko.make_cluster(f"shape{i}", B, R, 1, P, rf, rm, add, new_rf=new_rf).topics[0]
        if ko.provably_infeasible(ot):   # rigid bands on uneven racks: relax the rack bands, keep the rest
            bd = ot.bounds()
            ot.bounds_override = {"rack_lo": 0, "rack_hi": bd["rack_hi"] + P, "prack_hi": bd["prack_hi"] + 1, "rep_hi": bd["rep_hi"] + 1}
            assert not ko.provably_infeasible(ot), i
        _bound_chunk(monkeypatch, chunk)
        target = max(0, ko.upper_bound_simple(ot) - off)
        got = kao.dual_bound(to_product_topic(ot), target, iters=12, launches=2)
        st = kp.DualState(ot)
        for _ in range(2):
            st = kp.port_dual_bound(ot, target, 12, st)
            if st.flags & 7:
                break
        assert (got["iters"], got["flags"], got["best_dual"]) == (st.iters, st.flags, st.best_L), (i, got["iters"], st.iters, got["flags"], st.flags)
        assert got["a"].tolist() == st.a.tolist() and got["l"].tolist() == st.l.tolist() and got["g"].tolist() == st.g[:ot.n_racks].tolist(), i


def test_dual_bound_is_valid_and_closes_wide_family(kao, ko):
    """Every feasible instance of the wide family: floor(dual) never undercuts the HiGHS optimum and, aimed at the
    optimum, equals it on all but a few (the closed-form bound is tight on fewer than half)."""
    cases = _wide_cases(ko)
    closed = 0
    for c, ot in cases:
        got = kao.dual_bound(to_product_topic(ot), c["objective"], iters=1500)
        assert not got["flags"] & 4 and got["bound"] >= c["objective"], c["seed"]
        closed += got["bound"] == c["objective"]
    print("wide family: closed", closed, "of", len(cases))
    assert closed >= len(cases) - 1, (closed, len(cases))   # 174 of 175 (round 5, GPU call 36)


def test_solve_proves_wide_family(kao, ko):
    """kao_solve end to end on the 400-instance wide family in ONE call (heterogeneous topics): feasible instances
    reach the HiGHS optimum and are PROVEN optimal (closed-form bound or K-bound), every one of them (the schedule is
    deterministic: same seed, same answer); instances HiGHS found infeasible come back INFEASIBLE_PROVEN."""
    cases = load_golden("random_wide.json")["cases"]
    ots = [ko.random_case_wide(c["seed"]) for c in cases]
    res = kao.solve([to_product_topic(t) for t in ots], seed=31, restarts=32, iters_per_launch=256, time_limit_s=20.0, stop_at_bound=1)
    n_opt = n_proven = n_equal = 0
    for c, ot, r in zip(cases, ots, res):
        if c["status"] == "infeasible":
            assert r.status == "INFEASIBLE_PROVEN", c["seed"]
            continue
        n_opt += 1
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective <= c["objective"] <= r.upper_bound, (c["seed"], r.objective, c["objective"], r.upper_bound)
        n_equal += r.objective == c["objective"]
        if r.status == "OPTIMAL_PROVEN":
            assert r.objective == c["objective"], c["seed"]
            n_proven += 1
    assert n_equal == n_opt and n_proven == n_opt, (n_opt, n_equal, n_proven)   # every feasible instance: the exact optimum, proven


@pytest.mark.parametrize("name", ["cfg2_drift.json", "cfg3_drift.json", "cfg4_drift.json"])
def test_golden_optima_configs_drifted(kao, ko, name):
    """BASELINE configs 2-4 after a 20 % drift (config 4: bench.py's second time-to-optimal workload): HiGHS optimum
    reached and PROVEN -- the closed-form bound has a gap on each of these topics, so the proof is K-bound's."""
    g = load_golden(name)["topics"]
    ots = [ko.topic_from_dict(e["topic"]) for e in g]
    res = kao.solve([to_product_topic(t) for t in ots], seed=9, time_limit_s=20.0, stop_at_bound=1)
    for e, ot, r in zip(g, ots, res):
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective == e["objective"], (ot.name, r.objective, e["objective"])
        assert r.status == "OPTIMAL_PROVEN" and r.upper_bound == e["objective"] < e["upper_bound_closed_form"], (ot.name, r.status, r.upper_bound)


def test_dual_bound_limits_and_errors(kao, ko):
    ot = ko.readme_example()
    pt = to_product_topic(ot)
    got = kao.dual_bound(pt, 58, iters=200)
    assert got["bound"] == 58 and got["flags"] & 1                       # KAT-1: certificate == README optimum
    with pytest.raises(kao.KaoError):
        kao.dual_bound(pt, -1)
    big = to_product_topic(ko.make_cluster("big", 9000, 10, 1, 64, 3, [], []).topics[0])    # > 8192 brokers: outside K-bound
    with pytest.raises(kao.KaoError):
        kao.dual_bound(big, 10)
    with kao.Session([big, pt], restarts=8, iters_per_launch=64) as s:   # skipped silently inside a session, flag 8
        s.step(1)
        s.bound_step([100, 58], 100)
        b = s.bounds()
        assert b["flags"][0] == 8 and b["iters"][0] == 0 and b["upper_bound"][1] == 58


# ------------------------------------------------------------------------------- several GPUs in one process (kao_solve_multi)
def test_solve_multi_sharded_matches_single_device(kao, ko):
    """Topics dealt to two logical shards on device 0 through the C ABI: the same proven optima as kao_solve."""
    ots = ko.gen_config(4, n_topics=9).topics
    pts = [to_product_topic(t) for t in ots]
    one = kao.solve(pts, seed=5, time_limit_s=20)
    two = kao.solve_multi(pts, [0, 0], seed=5, time_limit_s=20)
    full = load_golden("cfg4_full.json")["topics"]
    for i, (ot, a, b) in enumerate(zip(ots, one, two)):
        assert a.status == b.status == "OPTIMAL_PROVEN"
        assert a.objective == b.objective == full[i]["objective"]
        obj, viol = ko.verify(ot, b.assignment)
        assert viol[0] == 0 and obj == b.objective


def test_solve_multi_replicated_exchanges_elites(kao, ko):
    """Fewer topics than devices: every (logical) device searches the topic, the best keys are min-reduced and the winner's
    assignment travels to the others (host copies for logical shards; one device alone goes through RCCL, world size 1)."""
    ot = _drifted(ko, 2, 1)[0]
    pt = to_product_topic(ot)
    g = load_golden("cfg2_drift.json")["topics"][0]
    for devices in ([0, 0, 0], [0]):
        r = kao.solve_multi([pt], devices, seed=21, time_limit_s=20, elite_period=2)[0]
        tm = kao.last_solve_timing()
        assert r.status == "OPTIMAL_PROVEN" and r.objective == g["objective"], (devices, r.status, r.objective)
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == r.objective


def test_solve_multi_replicated_runs_the_grouped_collectives(kao, ko, monkeypatch):
    """The RCCL control flow of the elite exchange with MORE THAN ONE rank (it had only ever run with a world of
    one): KAO_RCCL_LOOPBACK=1 serves ncclCommInitAll / ncclGroupStart / ncclAllReduce(ncclUint64, ncclMin) / ncclBroadcast /
    ncclGroupEnd from an in-process table, so three "ranks" on device 0 drive exactly the call sequence distinct GPUs would --
    one grouped all-reduce of the resident key buffers per exchange, one grouped broadcast per topic from the winner's rank."""
    monkeypatch.setenv("KAO_RCCL_LOOPBACK", "1")
    kao.rccl_selftest([0, 0, 0])                                  # both collectives, three communicators on one device
    ar0, bc0 = kao.rccl_loopback_counts()
    assert ar0 >= 1 and bc0 >= 1
    t = _drift_topic(100, 5, 1000)                                # 0.3-0.6 s to the proof: many launches with differing elites
    gold = [r for r in load_golden("drift_scale.json")["rows"] if (r["B"], r["P"]) == (100, 1000)][0]
    r = kao.solve_multi([t], [0, 0, 0], seed=21, time_limit_s=30, elite_period=2)[0]
    tm = kao.last_solve_timing()
    ar1, bc1 = kao.rccl_loopback_counts()
    assert r.status == "OPTIMAL_PROVEN" and r.objective == gold["milp_objective"], (r.status, r.objective)
    obj, viol = ko.verify(ko.Topic(name=t.name, broker_ids=np.array(t.broker_ids), rack_of=np.array(t.rack_of), n_racks=t.n_racks,
                                   n_partitions=t.n_partitions, rf=t.rf, current=np.array(t.current), weights=t.weights), r.assignment)
    assert viol[0] == 0 and obj == r.objective
    assert tm["elite_exchanges"] >= 1 and ar1 - ar0 == tm["elite_exchanges"] and 1 <= bc1 - bc0 <= tm["elite_exchanges"]
    # the same solve through plain copies (the hook off) gives the same answer: the collectives moved the right data
    monkeypatch.delenv("KAO_RCCL_LOOPBACK")
    r2 = kao.solve_multi([t], [0, 0, 0], seed=21, time_limit_s=30, elite_period=2)[0]
    assert (r2.status, r2.objective, r2.assignment.tolist()) == (r.status, r.objective, r.assignment.tolist())


def test_solve_multi_races_the_lp_on_a_replicated_large_topic(kao, ko, kp, monkeypatch):
    """Round 6: a replicated topic in the LP's regime (>= 32,768 replica slots) is a RACE -- every (logical) device runs the perturbed
    interior-point solve with its own salt, the first proof ends the solve for all, certificates and incumbents are shared through the
    grouped collectives as before.  Two ranks on device 0 (loop-back table): the proven optimum equals the single-device one, the answer
    satisfies every row under the scalar evaluator, and more than one LP was started."""
    from kafka_assignment_optimizer_amd import synthetic as sy
    monkeypatch.setenv("KAO_RCCL_LOOPBACK", "1")
    pt = sy.drift(sy.make_cluster(600, 12, 1, 12000, 3, [], []), 0.2, 1)[0]
    ot = _oracle_topic(ko, pt)
    one = kao.solve([pt], seed=5, stop_at_bound=1, time_limit_s=20.0)[0]
    two = kao.solve_multi([pt], [0, 0], seed=5, stop_at_bound=1, time_limit_s=20.0)[0]
    lp = kao.last_solve_lp()
    assert one.status == two.status == "OPTIMAL_PROVEN" and one.objective == two.objective == two.upper_bound, (one.status, two.status, one.objective, two.objective)
    obj, viol = kp.port_eval(ot, two.assignment)
    assert viol[0] == 0 and obj == two.objective
    assert lp["solves"] >= 1 and lp["adopted"] >= 1, lp


def test_solve_multi_shards_one_lp_over_the_devices(kao, ko, kp, monkeypatch):
    """Round 6, opt-in (KAO_MULTI_LP=shard): on a replicated topic in the LP's regime ONE interior-point solve runs sharded by partition
    range over all (logical) devices -- device 0's solve loop drives it through lp_open_fan, the shards' sums meet in grouped f64
    all-reduces (loop-back table here), the other devices' loops launch no K-search meanwhile.  Same proven optimum as one device, every
    row satisfied, and the collectives were really issued (hundreds of all-reduces: ~13 per iteration)."""
    from kafka_assignment_optimizer_amd import synthetic as sy
    monkeypatch.setenv("KAO_RCCL_LOOPBACK", "1")
    monkeypatch.setenv("KAO_MULTI_LP", "shard")
    pt = sy.drift(sy.make_cluster(600, 12, 1, 12000, 3, [], []), 0.2, 1)[0]
    ot = _oracle_topic(ko, pt)
    one = kao.solve([pt], seed=5, stop_at_bound=1, time_limit_s=20.0)[0]
    ar0, _ = kao.rccl_loopback_counts()
    two = kao.solve_multi([pt], [0, 0], seed=5, stop_at_bound=1, time_limit_s=30.0)[0]
    lp = kao.last_solve_lp()
    ar1, _ = kao.rccl_loopback_counts()
    print(f"sharded LP inside kao_solve_multi: {two.status} {two.objective} / {two.upper_bound}, {int(lp['iterations'])} iterations, {ar1 - ar0} all-reduces")
    assert one.status == two.status == "OPTIMAL_PROVEN" and one.objective == two.objective == two.upper_bound, (one.status, two.status, one.objective, two.objective)
    obj, viol = kp.port_eval(ot, two.assignment)
    assert viol[0] == 0 and obj == two.objective
    assert lp["solves"] >= 1 and lp["adopted"] >= 1 and ar1 - ar0 >= 10 * lp["iterations"], (lp, ar1 - ar0)


def test_rccl_collectives_on_the_resident_buffers(kao):
    """The RCCL path of kao_solve_multi (librccl.so opened on first use, ncclCommInitAll, ncclAllReduce(ncclUint64, ncclMin),
    ncclBroadcast) on the devices this box has -- one here, so a world of one; the same entry point checks 8 on a full node."""
    import torch
    kao.rccl_selftest(list(range(torch.cuda.device_count())))


def test_allreduce_best_resident_matches_host_packing(kao, ko):
    """multigpu.allreduce_best_resident (zero-copy view of the session's key buffer, packed and all-reduced on the GPU) ==
    multigpu.allreduce_best (host packing) -- RCCL world of one here."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
        import numpy as np, torch, torch.distributed as dist
        import kafka_assignment_optimizer_amd as kao
        from kafka_assignment_optimizer_amd import multigpu, synthetic
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(0); kao.init(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        topics = synthetic.drift(synthetic.make_config(4, n_topics=6), 0.2, 1)
        owned = [1, 3, 4, 6, 7, 9]
        with kao.Session(topics, seed=5, restarts=8, iters_per_launch=64) as s:
            s.step(2)
            a = multigpu.allreduce_best_resident(s, owned, 10, 0).cpu().numpy()
            b = multigpu.allreduce_best(s.best_keys(), owned, 10, 0, device=torch.device("cuda", 0))
        assert a.tolist() == b.tolist(), (a, b)
        assert (a[[0, 2, 5, 8]] == multigpu.KEY_NONE).all() and (a[owned] != multigpu.KEY_NONE).all()
        dist.destroy_process_group()
        print("resident-ok")
    """) % (ROOT_DIR, ROOT_DIR)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "resident-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_eval_counter_overflow_is_reported(kao, ko):
    """a candidate with more than 65,535 replicas on one broker would carry out of the 16-bit counter half --
    K-eval reports it (KAO_ERR_UNSUPPORTED) instead of returning wrong violation counts."""
    from kafka_assignment_optimizer_amd import Topic
    B, P = 8, 70000
    ok = (np.arange(P) % B).astype(np.uint16).reshape(P, 1)
    t = Topic(name="big", broker_ids=np.arange(B), rack_of=np.arange(B) % 2, n_racks=2, n_partitions=P, rf=1, current=ok.copy())
    obj, viol = kao.evaluate(t, ok)                      # 8750 per broker: fine
    assert viol[3] == 0
    bad = np.zeros((P, 1), dtype=np.uint16)              # 70,000 replicas on broker 0
    with pytest.raises(kao.KaoError) as e:
        kao.evaluate(t, bad)
    assert e.value.code == -2


# ------------------------------------------------------------------------------- replication factors 5..8 (two word groups per partition)
def _rf_cases(ko):
    return [(c, ko.random_case_rf(c["seed"])) for c in load_golden("random_rf.json")["cases"]]


def test_high_rf_eval_and_replay_bit_exact(kao, ko, kp):
    """K-eval (k_eval<8>) == the numpy verifier and K-search (k_search<*, *, 8>) == the scalar replay on topics with 5..8
    replicas, RF changes across the 4 / 5 boundary included (README.md:148-151: no cap on RF; README.md:9: RF changes)."""
    cases = [(c, t) for c, t in _rf_cases(ko) if c["status"] == "optimal"][:24]
    ots = [t for _, t in cases]
    assert max(t.rf for t in ots) == 8 and any(t.rf_cur <= 4 < t.rf for t in ots) and any(t.rf <= 4 < t.rf_cur for t in ots)
    for ot in ots[:10]:
        cands = random_candidates(ot, 12, seed=ot.n_partitions, p_mut=0.3, p_none=0.05)
        o, v = kao.evaluate_batch(to_product_topic(ot), cands)
        for i in range(len(cands)):
            oo, vv = ko.verify(ot, cands[i])
            assert (int(o[i]), v[i].tolist()) == (oo, vv.tolist()), (ot.name, i)
    seed = 808
    with kao.Session([to_product_topic(t) for t in ots], seed=seed, restarts=8, iters_per_launch=160) as s:
        s.step(2)
        assert s.stats()["drift"] == 0
        for ti, ot in enumerate(ots):
            for rho in (1, 6):
                dev = s.restart_state(ti, rho)
                ref = kp.port_search(ot, _tseed(seed, ti), rho, 2, 160)
                assert dev["final"].tolist() == ref["final"].tolist(), (ot.name, rho)
                assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])
                obj, viol = ko.verify(ot, dev["final"])
                assert (obj, int(viol[0])) == (dev["obj"], dev["V"])


def test_high_rf_golden_optima(kao, ko):
    """kao_solve on the whole RF 5..8 family: the HiGHS optimum on every feasible instance, every HiGHS-infeasible one either
    proven infeasible or left without a feasible plan, the canonical form of unique optima bit-exact."""
    cases = _rf_cases(ko)
    res = kao.solve([to_product_topic(t) for _, t in cases], seed=31, time_limit_s=30, max_launches=40)
    n_opt = n_proven = n_unique = 0
    for (c, ot), r in zip(cases, res):
        if c["status"] != "optimal":
            assert r.status in ("INFEASIBLE_PROVEN", "NO_FEASIBLE"), (c["seed"], r.status)
            continue
        assert r.objective == c["objective"], (c["seed"], r.objective, c["objective"], r.status)
        assert r.upper_bound >= c["objective"]
        obj, viol = ko.verify(ot, r.assignment)
        assert viol[0] == 0 and obj == c["objective"]
        n_opt += 1
        n_proven += r.status == "OPTIMAL_PROVEN"
        if c.get("unique"):
            assert kao.canonicalize(to_product_topic(ot), r.assignment).tolist() == ko.canonicalize(ot, np.array(c["assignment"])).tolist(), c["seed"]
            n_unique += 1
    print("high RF goldens: proven", n_proven, "of", n_opt)
    assert n_opt >= 50 and n_proven == n_opt, (n_opt, n_proven)   # 62 of 62 (round 5, GPU call 36)   # round 3: K-bound certifies RF 5..8 (it was n_opt // 3 on the closed-form bound)


# ------------------------------------------------------------------------------- broker weights and cluster-wide caps
def test_broker_weights_eval_and_replay_bit_exact(kao, ko, kp):
    """Broker weights (kao_topic.broker_w / broker_wl: plain extra coefficients of the `max:` row on every variable of a
    broker): K-eval == the verifier, K-search (priced instantiation, which carries the weight table) == the scalar replay."""
    ots = _drifted(ko, 4, 3) + [ko.random_case_rf(c["seed"]) for c in load_golden("random_rf.json")["cases"] if c["status"] == "optimal"][:2]
    rng = np.random.default_rng(12)
    for t in ots:
        t.broker_w = rng.integers(0, 6, t.n_brokers).astype(np.int32)
        t.broker_wl = rng.integers(0, 4, t.n_brokers).astype(np.int32)
    pts = [to_product_topic(t) for t in ots]
    for ot, pt in zip(ots, pts):
        cands = random_candidates(ot, 8, seed=3, p_mut=0.3, p_none=0.05)
        o, v = kao.evaluate_batch(pt, cands)
        for i in range(len(cands)):
            oo, vv = ko.verify(ot, cands[i])
            assert (int(o[i]), v[i].tolist()) == (oo, vv.tolist())
    seed = 515
    with kao.Session(pts, seed=seed, restarts=8, iters_per_launch=140) as s:
        s.step(2)
        assert s.stats()["drift"] == 0
        for ti, ot in enumerate(ots):
            for rho in (0, 7):
                dev = s.restart_state(ti, rho)
                ref = kp.port_search(ot, _tseed(seed, ti), rho, 2, 140)
                assert dev["final"].tolist() == ref["final"].tolist(), (ti, rho)
                assert (dev["best_obj"], dev["V"], dev["obj"], dev["n_accept"]) == (ref["best_obj"], ref["V"], ref["obj"], ref["n_accept"])
                obj, viol = ko.verify(ot, dev["final"])
                assert (obj, int(viol[0])) == (dev["obj"], dev["V"])
    # and the optimum of the weighted model (HiGHS) is what kao_solve returns; the certificate stays valid
    small = [ko.random_case(s, max_b=14, max_p=10) for s in range(40, 52)]
    for t in small:
        t.broker_w = rng.integers(0, 4, t.n_brokers).astype(np.int32)
        t.broker_wl = rng.integers(0, 3, t.n_brokers).astype(np.int32)
    res = kao.solve([to_product_topic(t) for t in small], seed=2, time_limit_s=10, max_launches=16)
    n = n_proven = 0
    for t, r in zip(small, res):
        ex = ko.solve_exact(t, 60)
        if ex.status != "optimal":
            continue
        assert r.objective == ex.objective <= r.upper_bound, (t.name, r.objective, ex.objective, r.upper_bound)
        n += 1
        n_proven += r.status == "OPTIMAL_PROVEN"
    print("broker-weight topics: proven", n_proven, "of", n)
    assert n >= 5 and n_proven == n, (n, n_proven)   # 10 of 10 (round 5, GPU call 36)   # round 3: K-bound prices weighted topics too


def test_solve_capped_matches_the_exact_joint_optimum_on_toys(kao, ko):
    """Cluster-wide per-broker caps (kao_solve_capped) against HiGHS on the JOINT model (every topic's README rows + one
    coupling row per capped broker over all topics; tests/golden/capped_toy.json): the plan respects every cap and every
    topic's own rows, its objective is sandwiched incumbent <= exact optimum <= Lagrangian bound, and it is within 2 % of the
    exact optimum (equal on most toys)."""
    cases = load_golden("capped_toy.json")["cases"]
    equal = 0
    for c in cases:
        ots = [ko.topic_from_dict(d) for d in c["topics"]]
        cap = np.array(c["replica_cap"])
        res, lb = kao.solve_capped([to_product_topic(t) for t in ots], cap, seed=c["seed"], time_limit_s=20, max_rounds=60)
        load = np.zeros(len(cap), dtype=int)
        total = 0
        for ot, r in zip(ots, res):
            assert r.status in ("FEASIBLE_BOUND_GAP", "OPTIMAL_PROVEN"), (c["seed"], r.status)
            obj, viol = ko.verify(ot, r.assignment)
            assert viol[0] == 0 and obj == r.objective
            total += obj
            np.add.at(load, r.assignment.reshape(-1).astype(int), 1)
        assert (load <= cap).all(), (c["seed"], load.tolist(), cap.tolist())
        assert total <= c["objective"] < c["objective_without_caps"]
        # round 3: K-bound prices weighted topics, so every round yields a Lagrangian bound -- and it meets the exact optimum
        assert lb is not None and 0 <= lb - c["objective"] <= 1, (c["seed"], lb, c["objective"])
        equal += total == c["objective"]
    assert equal == len(cases), equal   # round 3: 4 of 6 (the other two 2 units below); round 4 (KAO-CX slack nodes per rack): all six


def test_solve_capped_on_the_medium_golden(kao, ko):
    """Round 4: cluster-wide caps beyond toys -- 20 topics x 64 partitions on 60 brokers with 12 capped
    brokers, and 12 x 48 on 40 (tests/golden/capped_medium.json: exact joint optimum by HiGHS, make_golden_capped_medium.py).
    The plan respects every cap and every topic's own rows and is sandwiched plan <= exact optimum <= Lagrangian bound; round 4's
    first GPU run: 9165 / 9175 / 9195 and 4093 / 4097 / 4100 (plan / exact / bound) in 0.2 s; round 5: 9175 / 9175 / 9176 in 1.5 s and
    4097 / 4097 / 4097 in 0.7 s."""
    for c in load_golden("capped_medium.json")["cases"]:
        ots = [ko.topic_from_dict(d) for d in c["topics"]]
        cap = np.array(c["replica_cap"])
        res, lb = kao.solve_capped([to_product_topic(t) for t in ots], cap, seed=c["seed"], time_limit_s=20, max_rounds=60)
        load = np.zeros(len(cap), dtype=int)
        total = 0
        for ot, r in zip(ots, res):
            assert r.status in ("FEASIBLE_BOUND_GAP", "OPTIMAL_PROVEN"), (c["seed"], r.status)
            obj, viol = ko.verify(ot, r.assignment)
            assert viol[0] == 0 and obj == r.objective
            total += obj
            np.add.at(load, r.assignment.reshape(-1).astype(int), 1)
        assert (load <= cap).all(), (c["seed"], load.tolist(), cap.tolist())
        assert total <= c["objective"] < c["objective_without_caps"]
        # round 5 (a portfolio over the price granularity -- quarter, half, whole units -- with K-search's penalty range scaled with the
        # weights): plan = the exact joint optimum on both cases, bound one unit above it on the first, equal (OPTIMAL_PROVEN) on the second
        # (round 4, whole units only: 9169 / 9175 / 9195 and 4097 / 4097 / 4100)
        assert lb is not None and c["objective"] <= lb <= c["objective"] + 1, (c["seed"], lb, c["objective"])
        assert total == c["objective"], (c["seed"], total, c["objective"])
