#!/bin/bash
# round 5, call 32: KAO-LP's primal side on the device for the first time: perturbed LP + rounding against the certificate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c32
SALTS=0,1 timeout 900 python tools/r5_round_probe.py 300x6x2000 300x6x2000:2 270x6x2200 450x9x3500 500x10x5000 500x10x10000 1000x20x30000 > gpurun_out/${T}_round.log 2>&1
timeout 600 python - >> gpurun_out/${T}_round.log 2>&1 <<'P'
import sys, time
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
t = sy.north_star_topic('drift100k')
lb = kao.lp_bound(t)
for salt in (0, 1):
    t0 = time.perf_counter(); r = kao.lp_round(t, salt=salt); w = time.perf_counter() - t0
    print(f"1000x100000 salt {salt} pert {r['pert']:.2e}: objective {r['objective']} violations {r['violations'][0]} certificate {lb['bound']} | perturbed LP {r['iterations']} it status {r['status']} {r['ms_lp']:.0f} ms, rounding {r['ms_round']:.1f} ms, fractional {r['fractional']}, over inflow {r['over_inflow']}, wall {w:.2f} s", flush=True)
P
cat gpurun_out/${T}_round.log
