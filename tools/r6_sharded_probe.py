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
