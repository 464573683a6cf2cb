#!/bin/bash
# round 3, GPU call 22-23: re-fitted K-bound launch lengths; budget of the further starts by their gains: family (three solver seeds on the hard half) + scale
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
(time R3_HARD=1 R3_SEEDS=3,4,5 R3_SCHEDS=0 timeout 300 python tools/r3_probe.py family 3.0) > gpurun_out/r22_hard.log 2>&1
grep "proven" gpurun_out/r22_hard.log
(R3_SCHEDS=0 timeout 100 python tools/r3_probe.py scale 3.0) > gpurun_out/r22_scale.log 2>&1
cut -c1-200 gpurun_out/r22_scale.log
