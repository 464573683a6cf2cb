"""GPU (round 5): KAO-LP on the device against the scalar restatement (oracle/kao_lp_port.c): trace, LP value, certificate, time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
import kao_oracle as ko, kao_lp as kl
kao.init(0)
def otopic(pt):
    return ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                    n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights,
                    bounds_override=dict(pt.bounds_override))
args = sys.argv[1:] or ["100x5x1000", "130x5x1000", "450x9x3500"]
for arg in args:
    if "x" in arg and arg[0].isdigit():
        B, R, P = (int(x) for x in arg.split("x"))
        pt = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    else:                                   # a named north-star workload (synthetic.north_star_topic): drift100k, cfg5one, drift30k
        pt = sy.north_star_topic(arg)
        B, R, P = pt.n_brokers, pt.n_racks, pt.n_partitions
        kao.lp_trace(pt, max_iters=1)       # allocation warm-up
    t0 = time.perf_counter(); d = kao.lp_trace(pt); dt = time.perf_counter() - t0
    print(f"{B}x{P}: device status {d['status']} it {d['iterations']} primal {d['primal']:.6f} dual {d['dual']:.6f} ipm {d['ms']:.1f} ms (call {dt*1e3:.1f} ms)", flush=True)
    if P <= 6000:
        r = kl.port_solve(otopic(pt))
        print(f"   port   status {r['status']} it {r['iterations']} primal {r['primal']:.6f} dual {r['dual']:.6f}")
        n = min(len(d['trace']), len(r['trace']))
        worst = 0.0
        for i in range(n):
            a, b = d['trace'][i], r['trace'][i]
            rel = max(abs(a[0] - b[0]) / max(abs(b[0]), 1e-300), abs(a[1] - b[1]) / max(1.0, abs(b[1])), abs(a[2] - b[2]) / max(1.0, abs(b[2])))
            if b[0] > 1e-6: worst = max(worst, rel)
            if i < 3 or rel > 1e-6:
                print(f"   it {i}: mu {a[0]:.6e} / {b[0]:.6e}  pobj {a[1]:.6f} / {b[1]:.6f}  dobj {a[2]:.6f} / {b[2]:.6f}")
        print(f"   worst relative deviation of the trace while mu > 1e-6: {worst:.2e}; multipliers equal: {np.array_equal(d['a'], r['a']) and np.array_equal(d['l'], r['l']) and np.array_equal(d['g'], r['g'])}"
              f" (max |da| {np.abs(d['a'].astype(np.int64) - r['a']).max()})")
    t0 = time.perf_counter(); b = kao.lp_bound(pt); dt = time.perf_counter() - t0
    print(f"   lp_bound: certificate {b['bound']} (dual value {b['best_dual'] / 65536:.5f}), LP {b['dual']:.5f}, ipm {b['ms']:.1f} ms, whole call {dt*1e3:.1f} ms", flush=True)
