#!/bin/bash
# round 5, call 41: rounding v2 on the device's own iterates (north-star topic, three salts), LP tests, solve tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c41
timeout 600 python - > gpurun_out/${T}_round.log 2>&1 <<'P'
import sys, time
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for which in ('drift100k', 'drift30k'):
    t = sy.north_star_topic(which)
    for salt in (0, 1, 2):
        t0 = time.perf_counter(); r = kao.lp_round(t, salt=salt); w = time.perf_counter() - t0
        print(f"{which} salt {salt} pert {r['pert']:.2e}: objective {r['objective']} violations {r['violations']} | {r['iterations']} it status {r['status']} {r['ms_lp']:.0f} ms, rounding {r['ms_round']:.1f} ms, fractional {r['fractional']}, over inflow {r['over_inflow']}, wall {w:.2f} s", flush=True)
P
cat gpurun_out/${T}_round.log | cut -c1-250
SALTS=0,1,2 timeout 600 python tools/r5_round_probe.py 500x10x5000 500x10x10000 450x9x3500 >> gpurun_out/${T}_round.log 2>&1; tail -9 gpurun_out/${T}_round.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_lp.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|golden families"
