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
