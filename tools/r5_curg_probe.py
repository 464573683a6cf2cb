"""GPU (round 5): K-search launch time of a topic between 4,900 and 9,800 partitions: working words in LDS (k_search_curg) against
the HBM path (KAO_CUR_GLOBAL=0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for B, R, P in ((500, 10, 5000), (500, 10, 9000)):
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    for restarts in (256, 1024):
        with kao.Session([t], seed=3, restarts=restarts, iters_per_launch=512, profile=1) as s:
            s.step(1); s.sync(); a = s.stats()
            t0 = time.perf_counter(); s.step(6); s.sync(); wall = time.perf_counter() - t0
            b = s.stats()
            r = s.best()[0]
        print(f"{B}x{P} restarts {restarts} KAO_CUR_GLOBAL={os.environ.get('KAO_CUR_GLOBAL', '1')}: k_search {(b['ms_search'] - a['ms_search']) / 6:.3f} ms/launch, "
              f"wall {1e3 * wall / 6:.3f} ms/step, LDS {b['lds_bytes_search']} B, workgroups {b['blocks_search']}, objective after 7 launches {r.objective} (viol {r.violations[0]})", flush=True)
