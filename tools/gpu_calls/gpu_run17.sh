#!/bin/bash
# round 3, GPU call 17: adaptive budget of the further KAO-CX starts: family + scale probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r17_family.log 2>&1
grep "family sched.: proven" gpurun_out/r17_family.log; grep "scale" gpurun_out/r17_family.log | cut -c1-220
