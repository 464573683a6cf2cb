#!/bin/bash
# round 5, call 39: north-star solves under limits of 3 / 2 / 1.5 / 1 s with the LP driven a few marks ahead; determinism + LP solve tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c39
timeout 600 python - > gpurun_out/${T}_solve.log 2>&1 <<'P'
import sys, time
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.north_star_topic('drift100k')
kao.solve([t], seed=1, max_launches=1)
for budget in (3.0, 2.0, 1.5, 1.0):
    t0 = time.perf_counter(); r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]; dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print(f"drift100k limit {budget}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} t_best {tm['time_to_best']:.3f}s read back {tm['results_read_back']:.3f}s total {dt:.3f}s launches {tm['launches']} cx {tm['cx_calls']} lp {lp}", flush=True)
t = sy.north_star_topic('drift30k')
for budget in (3.0,):
    t0 = time.perf_counter(); r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]; dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print(f"drift30k limit {budget}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} t_best {tm['time_to_best']:.3f}s read back {tm['results_read_back']:.3f}s total {dt:.3f}s launches {tm['launches']} cx {tm['cx_calls']} lp {lp}", flush=True)
P
cat gpurun_out/${T}_solve.log | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lp.py -m gpu -q -k "deterministic or north_star or solve_proves or retries" 2>&1 | tail -3
