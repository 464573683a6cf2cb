"""Time-to-bound on a hard single topic: 3000 partitions x RF 3 on 1000 brokers / 20 racks, 3 brokers replaced;
every band is an equality (9 replicas and 3 leaders per broker, 450 per rack)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic

kao.init(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
t = synthetic.make_cluster(1000, 20, 1, P, 3, [7, 77, 777], [(1000, 7), (1001, 17), (1002, 17)])[0]
print("bounds", kao.derive_bounds(t), "upper bound", kao.upper_bound(t))
for iters in (64, 256):
    t0 = time.perf_counter()
    r = kao.solve([t], seed=3, iters_per_launch=iters, stop_at_bound=1, time_limit_s=10.0)[0]
    dt = time.perf_counter() - t0
    tm = kao.last_solve_timing()
    print(f"iters/launch {iters}: status {r.status} objective {r.objective} / {r.upper_bound} viol {r.violations[0]} "
          f"launches {tm['launches']} time_to_best {tm['time_to_best']:.4f}s total {dt:.3f}s")
