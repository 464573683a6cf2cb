"""The product-side synthetic generator (bench.py's input) equals the oracle's independent one."""
import numpy as np
import pytest


@pytest.mark.parametrize("cfg,k", [(2, 1), (3, 3), (4, 4), (5, 2)])
def test_generators_agree(ko, cfg, k):
    from kafka_assignment_optimizer_amd import synthetic
    a = synthetic.make_config(cfg, n_topics=k)
    b = ko.gen_config(cfg, n_topics=k).topics
    assert len(a) == len(b) == k
    for x, y in zip(a, b):
        assert x.name == y.name and x.n_racks == y.n_racks and x.rf == y.rf
        assert np.array_equal(x.broker_ids, y.broker_ids) and np.array_equal(x.rack_of, y.rack_of)
        assert np.array_equal(x.current, y.current)
        assert dict(x.bounds_override) == dict(y.bounds_override)


def test_drift_is_deterministic_and_matches_the_golden_instances(ko):
    """synthetic.drift (bench.py's drifted workload): reproducible, keeps the bands / brokers / racks, changes about the
    stated fraction of slots, never duplicates a broker inside a partition, and regenerates exactly the instances whose
    HiGHS optima are stored in tests/golden/cfg4_drift.json."""
    from conftest import load_golden
    from kafka_assignment_optimizer_amd import synthetic
    base = synthetic.make_config(4, n_topics=4)
    a = synthetic.drift(base, 0.2, 1)
    b = synthetic.drift(base, 0.2, 1)
    g = load_golden("cfg4_drift.json")["topics"]
    for x, y, o, e in zip(a, b, base, g):
        assert np.array_equal(x.current, y.current)
        assert np.array_equal(x.broker_ids, o.broker_ids) and np.array_equal(x.rack_of, o.rack_of) and x.rf == o.rf
        changed = int((x.current != o.current).sum())
        assert 0.1 * x.current.size <= changed <= 0.2 * x.current.size
        for row in x.current:
            vals = [v for v in row.tolist() if v != 0xFFFF]
            assert len(vals) == len(set(vals))
        assert np.array_equal(x.current, np.array(e["topic"]["current"], dtype=np.uint16))
        assert e["objective"] < e["upper_bound_closed_form"]   # the closed-form bound has a gap on every one of them
    assert not np.array_equal(a[0].current, synthetic.drift(base, 0.2, 2)[0].current)
