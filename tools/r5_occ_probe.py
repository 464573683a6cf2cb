"""GPU (round 5): K-search launch time by instantiation (plain / priced, 4 / 8 words, small / wide topics in LDS) -- what the
occupancy floor of the register allocator (amdgpu_waves_per_eu on k_search) buys or costs outside the bench workload."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
tag = os.environ.get("TAG", "")
shapes = [   # name, brokers, racks, topics, partitions, rf, restarts, priced
    ("small plain   500x50  rf3 x200", 500, 10, 200, 50, 3, 160, False),
    ("small priced  500x50  rf3 x200", 500, 10, 200, 50, 3, 160, True),
    ("small plain   500x50  rf5 x200", 500, 10, 200, 50, 5, 160, False),
    ("small priced  500x50  rf5 x200", 500, 10, 200, 50, 5, 160, True),
    ("wide  plain   100x256 rf3 x50 ", 100, 4, 50, 256, 3, 256, False),
    ("wide  priced  100x256 rf3 x50 ", 100, 4, 50, 256, 3, 256, True),
    ("wide  plain   200x640 rf3 x20 ", 200, 6, 20, 640, 3, 512, False),
    ("wide  priced  300x2000 rf3 x1 ", 300, 6, 1, 2000, 3, 1024, True),
    ("wide  plain   300x2000 rf5 x1 ", 300, 6, 1, 2000, 5, 1024, False),
]
for name, B, R, T, P, rf, restarts, priced in shapes:
    ts = sy.drift(sy.make_cluster(B, R, T, P, rf, [], []), 0.2, 1)
    with kao.Session(ts, seed=3, restarts=restarts, iters_per_launch=512, profile=1) as s:
        if priced:
            rng = np.random.default_rng(5)
            for i, t in enumerate(ts):
                s.set_prices(i, rng.integers(-2, 3, t.n_brokers) * 16384, rng.integers(-2, 3, t.n_brokers) * 16384, rng.integers(-1, 2, t.n_racks) * 16384)
        s.step(2); s.sync(); a = s.stats()
        s.step(6); s.sync(); b = s.stats()
        obj = sum(r.objective for r in s.best())
    print(f"{tag:6s} {name}: k_search {(b['ms_search'] - a['ms_search']) / 6:8.3f} ms/launch  LDS {b['lds_bytes_search']:6d} B  workgroups {b['blocks_search']:6d}  objective sum {obj}", flush=True)
