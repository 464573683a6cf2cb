"""GPU (round 6): ONE topic's LP over logical shards of one device (kao_lp_sharded_test, KAO_RCCL_LOOPBACK=1) against the whole-topic
solve: certificate, rounded objective, iterations, collectives.  (Timing says nothing: the shards share one GPU and the loop-back
collectives are host-synchronous copies.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KAO_RCCL_LOOPBACK"] = "1"
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for name, shard_counts in (("drift5k", (1, 2, 3)), ("drift30k", (2, 4)), ("drift100k", (2, 8))):
    pt = sy.north_star_topic(name)
    eps = min(1e-4, 1.5 / (pt.n_partitions * pt.rf))
    cert = kao.lp_bound(pt)["bound"]
    whole = kao.lp_round(pt, pert=eps, tol=1e-10, max_iters=200)
    print(f"{name}: whole topic: certificate {cert}, rounded {whole['objective']} ({whole['violations'][0]} violations), {whole['iterations']} iterations, {whole['ms_lp']:.0f} ms", flush=True)
    for n in shard_counts:
        t0 = time.perf_counter()
        sh = kao.lp_sharded(pt, [0] * n, pert=eps, tol=1e-10, max_iters=200)
        print(f"   {n} shards: certificate {sh['bound']}, rounded {sh['objective']} ({sh['violations'][0]} violations), {sh['iterations']} iterations (status {sh['status']}), "
              f"{sh['collectives']} collectives = {sh['collectives'] / max(1, sh['iterations']):.1f} per iteration, {sh['ms_lp']:.0f} ms (wall {time.perf_counter() - t0:.1f} s)", flush=True)

# the same inside kao_solve_multi (opt-in KAO_MULTI_LP=shard): device 0's solve loop drives ONE sharded LP over all logical devices
os.environ["KAO_MULTI_LP"] = "shard"
for name, n in (("drift30k", 2), ("drift100k", 2), ("drift100k", 4)):
    pt = sy.north_star_topic(name)
    ar0, _ = kao.rccl_loopback_counts()
    t0 = time.perf_counter()
    r = kao.solve_multi([pt], [0] * n, seed=3, stop_at_bound=1, time_limit_s=60.0)[0]
    lp = kao.last_solve_lp(); ar1, _ = kao.rccl_loopback_counts()
    print(f"kao_solve_multi {name} on {n} logical devices, KAO_MULTI_LP=shard: {r.status} objective {r.objective} certificate {r.upper_bound}, {int(lp['solves'])} LP solve(s), "
          f"{int(lp['iterations'])} iterations, {ar1 - ar0} all-reduces, wall {time.perf_counter() - t0:.1f} s (loop-back collectives: host-synchronous)", flush=True)
