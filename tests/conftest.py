import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU (sanitiser builds); runs only when selected with -m slow")


def pytest_collection_modifyitems(config, items):
    if "slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="slow: select with -m slow")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def to_product_topic(ot):
    """oracle Topic -> product Topic (the two classes are deliberately independent)."""
    from kafka_assignment_optimizer_amd import Topic
    return Topic(name=ot.name, broker_ids=np.array(ot.broker_ids), rack_of=np.array(ot.rack_of), n_racks=ot.n_racks,
                 n_partitions=ot.n_partitions, rf=ot.rf, current=np.array(ot.current), weights=ot.weights,
                 partition_ids=ot.partition_ids, bounds_override=dict(ot.bounds_override),
                 broker_w=None if getattr(ot, "broker_w", None) is None else np.array(ot.broker_w),
                 broker_wl=None if getattr(ot, "broker_wl", None) is None else np.array(ot.broker_wl))


def random_candidates(ot, n, seed, p_mut=0.15, p_none=0.02):
    """n compact candidates: the current assignment (holes filled randomly) with random mutations,
    including out-of-range / empty slots and duplicates, so every violation family is exercised."""
    rng = np.random.default_rng(seed)
    P, RF, B = ot.n_partitions, ot.rf, ot.n_brokers
    base = np.full((P, RF), 0xFFFF, dtype=np.int64)
    k = min(RF, ot.rf_cur)
    base[:, :k] = np.asarray(ot.current[:, :k], dtype=np.int64)
    out = np.repeat(base[None], n, axis=0)
    holes = out == 0xFFFF
    out[holes] = rng.integers(0, B, size=int(holes.sum()))
    mut = rng.random(out.shape) < p_mut
    out[mut] = rng.integers(0, B, size=int(mut.sum()))
    none = rng.random(out.shape) < p_none
    out[none] = rng.choice(np.array([0xFFFF, B, B + 7]), size=int(none.sum()))
    out[0] = base  # one candidate keeps the raw current assignment (holes as NONE)
    return out.astype(np.uint16)


@pytest.fixture(scope="session")
def ko():
    import kao_oracle
    return kao_oracle


@pytest.fixture(scope="session")
def kp():
    import kao_port
    kao_port.build()
    return kao_port
