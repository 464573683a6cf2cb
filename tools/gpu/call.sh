#!/bin/bash
# round 5, call 19: the compound-edge layer on the elite's KAO-CX fixpoints (KAO_CX_PAIRS=1) now that the certificate is there early: hard half of the family, the d2 golden, slack-band and 5,000-partition topics
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c19
for PAIRS in 1; do
(time KAO_CX_PAIRS=$PAIRS R3_HARD=1 R3_SCHEDS=0 R3_SEEDS=3,4,5 timeout 600 python tools/r3_probe.py family 3) > gpurun_out/${T}_family_pairs${PAIRS}.log 2>&1
grep "proven\|real" gpurun_out/${T}_family_pairs${PAIRS}.log | cut -c1-250
(time KAO_CX_PAIRS=$PAIRS DSEED=2 SEEDS=1,2,3,4,5 BUDGET=6 timeout 300 python tools/r5_solve_probe.py 300x6x2000) > gpurun_out/${T}_d2_pairs${PAIRS}.log 2>&1
cut -c1-120 gpurun_out/${T}_d2_pairs${PAIRS}.log
(time KAO_CX_PAIRS=$PAIRS SEEDS=3,4,5 BUDGET=4 timeout 300 python tools/r5_solve_probe.py 270x6x2200 500x10x5000) > gpurun_out/${T}_mid_pairs${PAIRS}.log 2>&1
cut -c1-120 gpurun_out/${T}_mid_pairs${PAIRS}.log
done
