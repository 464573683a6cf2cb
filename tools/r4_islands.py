"""GPU probe of round 4 (measurement tooling): would K independent populations of ONE topic ("islands": the topic handed to
kao_solve K times -- every copy gets its own seed, its own K-bound, its own KAO-CX calls) prove more of the hard half of the
drifted family inside the same wall-clock budget than one population does?  A single topic of 2,000 partitions occupies 40 of
256 compute units, so the copies search for free; what they cost is host-side KAO-CX time.
usage: r4_islands.py "<K,...>" "<solver seed,...>" [budget_s]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy

kao.init(0)
FAMILY = [(200, 8, 1600), (200, 5, 2000), (250, 10, 2000), (300, 6, 2000), (160, 4, 2400), (400, 8, 3000), (250, 5, 4000)]
Ks = [int(v) for v in sys.argv[1].split(",")]
seeds = [int(v) for v in sys.argv[2].split(",")]
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
for K in Ks:
    for sd in seeds:
        proven, gaps = 0, []
        for (B, R, P) in FAMILY:
            for dseed in (1, 2):
                t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, dseed)[0]
                t0 = time.perf_counter()
                rs = kao.solve([t] * K, seed=sd, stop_at_bound=1, time_limit_s=budget)
                dt = time.perf_counter() - t0
                tm = kao.last_solve_timing()
                best = max(int(r.objective) for r in rs)
                cert = min(int(r.upper_bound) for r in rs)
                ok = best >= cert
                proven += ok
                gaps.append(cert - best)
                print(f"islands K={K} seed {sd} B={B:4d} R={R:2d} P={P:5d} d{dseed}: best {best} cert {cert} gap {cert - best} {dt:.2f}s "
                      f"copies {[int(r.objective) for r in rs]} launches {tm['launches']} cx {tm['cx_calls']}/{tm['cx_gains']} gens {tm['generations']}", flush=True)
        print(f"islands K={K} seed {sd}: proven {proven}/{len(gaps)}; gaps of the others {sorted(g for g in gaps if g)}", flush=True)
