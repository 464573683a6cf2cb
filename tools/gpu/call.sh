#!/bin/bash
# round 5, call 37: LP tests with the primal side, the fixed / tightened solve tests, LP time after the split dot products in the diagonal tile, the north-star solve with the LP alone from the start
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c37
timeout 1500 python -m pytest tests/test_gpu_lp.py -m gpu -q -s 2>&1 | grep -v "^\[kao" > gpurun_out/${T}_pytest.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cycle.py -m gpu -q -s -k "north_star or further_kao_cx or beats_plain or deterministic or wide_family or high_rf_golden or broker_weights" 2>&1 | grep -v "^\[kao" >> gpurun_out/${T}_pytest.log
grep -E "^FAILED|^ERROR|passed|failed|golden families|drifted 1000" gpurun_out/${T}_pytest.log | cut -c1-250
timeout 600 python - > gpurun_out/${T}_lp.log 2>&1 <<'P'
import sys, time
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
for budget in (3.0, 2.0):
    t0 = time.perf_counter(); r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]; dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print(f"drift100k limit {budget}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} t_best {tm['time_to_best']:.3f}s read back {tm['results_read_back']:.3f}s total {dt:.3f}s launches {tm['launches']} cx {tm['cx_calls']} lp {lp}", flush=True)
P
cat gpurun_out/${T}_lp.log | cut -c1-300
