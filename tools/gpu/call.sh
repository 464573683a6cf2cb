#!/bin/bash
# round 5, call 49: what the rounded iterates of 2000 x 100,000 and of the 40 % drift violate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c49
timeout 900 python - > gpurun_out/${T}_viol.log 2>&1 <<'P'
import sys, time, os
import numpy as np
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for name, t in (('2000x100000', sy.drift(sy.make_cluster(2000, 20, 1, 100000, 3, [], []), 0.2, 1)[0]), ('1000x100000 drift 0.4', sy.drift(sy.make_cluster(1000, 20, 1, 100000, 3, [], []), 0.4, 1)[0])):
    b = kao.lp_bound(t)
    print(name, 'certificate', b['bound'], b['iterations'], 'it', round(b['ms']), 'ms', flush=True)
    for pert, salt, tol in ((1.5 / (t.n_partitions * 3), 0, 1e-10), (0.0, 1, 0.0)):
        r = kao.lp_round(t, pert=pert, salt=salt, tol=tol, max_iters=200)
        A = r['assignment']
        B = t.n_brokers
        load = np.bincount(A.reshape(-1), minlength=B); lead = np.bincount(A[:, 0], minlength=B)
        print(f"  pert {r['pert']:.1e} salt {salt}: objective {r['objective']} violations {r['violations']} | {r['iterations']} it status {r['status']} {r['ms_lp']:.0f} ms, rounding {r['ms_round']:.1f} ms, fractional {r['fractional']}, over inflow {r['over_inflow']}; replica loads min {load.min()} max {load.max()} leaders min {lead.min()} max {lead.max()}", flush=True)
P
cat gpurun_out/${T}_viol.log | cut -c1-300
