#!/bin/bash
# round 5, call 51: 2000 x 100,000 with the chain repair: what the rounded iterates violate now, and the solve
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c51
timeout 900 python - > gpurun_out/${T}_2000.log 2>&1 <<'P'
import sys, time, os
import numpy as np
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.drift(sy.make_cluster(2000, 20, 1, 100000, 3, [], []), 0.2, 1)[0]
for pert, salt, tol in ((0.0, 1, 0.0), (0.0, 2, 0.0)):
    r = kao.lp_round(t, pert=pert, salt=salt, tol=tol, max_iters=200)
    A = r['assignment']; B = t.n_brokers
    load = np.bincount(A.reshape(-1), minlength=B); lead = np.bincount(A[:, 0], minlength=B)
    print(f"2000x100000 pert {r['pert']:.1e} salt {salt}: objective {r['objective']} violations {r['violations']} | {r['iterations']} it status {r['status']} {r['ms_lp']:.0f} ms, rounding {r['ms_round']:.1f} ms, fractional {r['fractional']}; replica loads min {load.min()} max {load.max()} leaders min {lead.min()} max {lead.max()}", flush=True)
kao.solve([t], seed=1, max_launches=1)
t0 = time.perf_counter(); r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=8.0)[0]; dt = time.perf_counter() - t0
tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
print(f"2000x100000 solve: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} read back {tm['results_read_back']:.3f}s launches {tm['launches']} cx {tm['cx_calls']} lp {lp}", flush=True)
P
cat gpurun_out/${T}_2000.log | cut -c1-300
