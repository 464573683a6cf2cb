"""GPU: K-bound microseconds per iteration on large drifted topics for the three drivers (test tooling): `one` = k_bound's single
persistent workgroup, `step:<chunk>` = k_bound_step (one kernel launch per iteration, round 2), `multi:<chunk>` = k_bound_multi
(persistent, one barrier per iteration, round 3).  The rate is the slope between two launch lengths, so session set-up cancels;
the best dual value is printed so that the drivers can be seen to agree."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
shapes = ((100, 5, 1000), (300, 6, 2000), (500, 10, 5000), (500, 10, 10000), (1000, 20, 30000))
if os.environ.get("BOUND_RATE_SHAPES"):
    shapes = tuple(tuple(int(v) for v in x.split("x")) for x in os.environ["BOUND_RATE_SHAPES"].split(","))
specs = sys.argv[1:] or ["one", "step:512", "multi:512:8", "multi:512:16", "multi:1024:16", "multi:256:16"]
for B, R, P in shapes:
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    target = int(kao.upper_bound(t) * 0.9)     # far below: no launch stops early
    row = []
    for spec in specs:
        drv, _, ch = spec.partition(":")
        ch, _, wv = ch.partition(":")
        os.environ.pop("KAO_BOUND_WAVES", None)
        if wv:
            os.environ["KAO_BOUND_WAVES"] = wv
        os.environ["KAO_BOUND_MULTI"] = "1" if drv == "multi" else "0"
        os.environ["KAO_BOUND_CHUNK"] = "0" if drv == "one" else ch
        n1, n2 = (20, 120) if drv == "one" and P >= 10000 else ((100, 600) if drv != "multi" else (500, 4500))
        ts = []
        for n in (n1, n2):
            t0 = time.perf_counter()
            got = kao.dual_bound(t, target, iters=n, launches=1)
            ts.append(time.perf_counter() - t0)
            assert got["iters"] == n, (spec, got["iters"], got["flags"])
        us = (ts[1] - ts[0]) / (n2 - n1) * 1e6
        row.append(f"{spec}: {us:.1f} us/iter (dual {got['best_dual']}, flags {got['flags']})")
    print(f"B={B} P={P}: " + "; ".join(row), flush=True)
