#!/bin/bash
# round 5, call 53: two centrality correctors per interior-point iteration on the device: LP tests (trace against the restatement), iterations and time, solves
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c53
timeout 600 python -m pytest tests/test_gpu_lp.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -8
timeout 300 python - > gpurun_out/${T}_mcc.log 2>&1 <<'P'
import sys, time, os
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.north_star_topic('drift100k')
kao.lp_trace(t, max_iters=1)
b = kao.lp_bound(t)
print('drift100k certificate', b['bound'], b['iterations'], 'it', round(b['ms'], 1), 'ms =', round(b['ms'] / b['iterations'], 2), 'ms / it', flush=True)
kao.solve([t], seed=1, max_launches=1)
cases = [('drift100k', t), ('100k seed 2', sy.drift(sy.make_cluster(1000, 20, 1, 100000, 3, [], []), 0.2, 2)[0]), ('drift30k', sy.north_star_topic('drift30k')), ('500x5000', sy.drift(sy.make_cluster(500, 10, 1, 5000, 3, [], []), 0.2, 1)[0])]
for name, tt in cases:
    t0 = time.perf_counter(); r = kao.solve([tt], seed=3, stop_at_bound=1, time_limit_s=4.0)[0]; dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print(f"{name}: {r.status} objective {r.objective} certificate {r.upper_bound} read back {tm['results_read_back']:.3f}s cx {tm['cx_calls']} lp {lp}", flush=True)
P
cat gpurun_out/${T}_mcc.log | cut -c1-250
