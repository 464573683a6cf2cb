#!/bin/bash
# round 5, call 34: kao_solve with the perturbed LP (certificate + rounded iterate from one solve)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c34
SEEDS=1,2,3 BUDGET=4 timeout 900 python tools/r5_solve_probe.py 300x6x2000 270x6x2200 450x9x3500 500x10x5000 500x10x10000 1000x20x30000 > gpurun_out/${T}_solve.log 2>&1
DSEED=2 SEEDS=1,2,3,4,5 BUDGET=4 timeout 600 python tools/r5_solve_probe.py 300x6x2000 >> gpurun_out/${T}_solve.log 2>&1
KAO_SOLVE_TRACE=1 timeout 600 python - >> gpurun_out/${T}_solve.log 2> gpurun_out/${T}_trace100k.log <<'P'
import sys, time
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.north_star_topic('drift100k')
kao.solve([t], seed=1, max_launches=1)
for budget in (3.0, 1.0):
    t0 = time.perf_counter(); r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]; dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print(f"drift100k limit {budget}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} t_best {tm['time_to_best']:.3f}s total {dt:.3f}s launches {tm['launches']} cx {tm['cx_calls']} lp {lp}", flush=True)
P
grep -v "^\[kao-solve\] launch\|generation" gpurun_out/${T}_trace100k.log | grep "KAO-LP\|KAO-CX" | head -20 >> gpurun_out/${T}_solve.log
cat gpurun_out/${T}_solve.log | cut -c1-330
