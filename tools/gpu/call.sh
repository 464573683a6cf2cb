#!/bin/bash
# round 5, call 2: KAO-LP inside kao_solve (iterations beside the K-search launches, device-resident scalars) -- the stalled topics of round 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c02
(time timeout 300 python tools/r5_lp_probe.py 130x5x1000 450x9x3500 1000x20x30000) > gpurun_out/${T}_lp_probe.log 2>&1
tail -30 gpurun_out/${T}_lp_probe.log | cut -c1-250
(time SEEDS=3,4,5 BUDGET=4 timeout 600 python tools/r5_solve_probe.py 270x6x2200 350x7x2500 450x9x3500 400x8x3000 500x10x5000 500x10x10000 1000x20x30000) > gpurun_out/${T}_solve.log 2>&1
cat gpurun_out/${T}_solve.log | cut -c1-300
(time DSEED=2 SEEDS=1,2,3,4,5 BUDGET=8 timeout 300 python tools/r5_solve_probe.py 300x6x2000) > gpurun_out/${T}_solve_d2.log 2>&1
cat gpurun_out/${T}_solve_d2.log | cut -c1-300
(time KAO_SOLVE_TRACE=1 SEEDS=3 BUDGET=3 timeout 100 python tools/r5_solve_probe.py 450x9x3500) > gpurun_out/${T}_trace450.log 2>&1
grep -n "KAO-LP" gpurun_out/${T}_trace450.log | head
