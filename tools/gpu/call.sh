#!/bin/bash
# round 5, call 42: rounding with the two-pass completion on the device's iterates; a half-integral first iterate inside kao_solve (retry path) at 100,000 partitions
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c42
timeout 600 python - > gpurun_out/${T}_round.log 2>&1 <<'P'
import sys, time, os
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.north_star_topic('drift100k')
for salt in (0, 1, 2, 3):
    t0 = time.perf_counter(); r = kao.lp_round(t, salt=salt); w = time.perf_counter() - t0
    print(f"drift100k salt {salt} pert {r['pert']:.2e}: objective {r['objective']} violations {r['violations']} | {r['iterations']} it status {r['status']} {r['ms_lp']:.0f} ms, rounding {r['ms_round']:.1f} ms, fractional {r['fractional']}, over inflow {r['over_inflow']}, wall {w:.2f} s", flush=True)
kao.solve([t], seed=1, max_launches=1)
os.environ['KAO_LP_PERT'] = '3.33e-4'     # the perturbation whose salt-0 iterate is half-integral in 20 partitions
for budget in (4.0,):
    t0 = time.perf_counter(); r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget)[0]; dt = time.perf_counter() - t0
    tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
    print(f"drift100k KAO_LP_PERT=3.33e-4 limit {budget}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} t_best {tm['time_to_best']:.3f}s read back {tm['results_read_back']:.3f}s launches {tm['launches']} cx {tm['cx_calls']} lp {lp}", flush=True)
del os.environ['KAO_LP_PERT']
P
cat gpurun_out/${T}_round.log | cut -c1-300
SALTS=0,1,2,3 timeout 600 python tools/r5_round_probe.py 500x10x5000 >> gpurun_out/${T}_round.log 2>&1; tail -4 gpurun_out/${T}_round.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_lp.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|golden families"
