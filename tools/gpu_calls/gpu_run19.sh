#!/bin/bash
# round 3, GPU call 19: further KAO-CX starts against none, hard half of the family, three solver seeds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
(time R3_HARD=1 R3_SEEDS=3,4,5 KAO_DET_CX_STARTS=0 R3_SCHEDS=0 timeout 300 python tools/r3_probe.py family 3.0) > gpurun_out/r19_hard_s0.log 2>&1
grep "proven" gpurun_out/r19_hard_s0.log
(time R3_HARD=1 R3_SEEDS=3,4,5 R3_SCHEDS=0 timeout 300 python tools/r3_probe.py family 3.0) > gpurun_out/r19_hard_adapt.log 2>&1
grep "proven" gpurun_out/r19_hard_adapt.log
