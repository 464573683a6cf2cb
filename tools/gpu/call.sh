#!/bin/bash
# round 5, call 52: k_lp_schur_broker as straight-line code over four incidences: LP tests, time per iteration, the north-star solve
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c52
timeout 900 python -m pytest tests/test_gpu_lp.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
timeout 600 python - > gpurun_out/${T}_lp.log 2>&1 <<'P'
import sys, time, os
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for which in ('drift30k', 'drift100k'):
    t = sy.north_star_topic(which)
    kao.lp_trace(t, max_iters=1)
    b = kao.lp_bound(t)
    print(which, 'certificate', b['bound'], b['iterations'], 'it', round(b['ms'], 1), 'ms =', round(b['ms'] / b['iterations'], 2), 'ms / it', flush=True)
t = sy.north_star_topic('drift100k')
kao.solve([t], seed=1, max_launches=1)
for budget in (3.0, 1.5):
    t0 = time.perf_counter(); r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]; dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print(f"drift100k limit {budget}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} read back {tm['results_read_back']:.3f}s total {dt:.3f}s lp {lp}", flush=True)
P
cat gpurun_out/${T}_lp.log | cut -c1-260
