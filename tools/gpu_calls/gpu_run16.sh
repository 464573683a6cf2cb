#!/bin/bash
# round 3, GPU call 16: further KAO-CX starting points (other restarts' best snapshots): whole GPU suite, family probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -q -x) > gpurun_out/r16_pytest.log 2>&1
tail -5 gpurun_out/r16_pytest.log
(time R3_SCHEDS=0 timeout 300 python tools/r3_probe.py family 3.0) > gpurun_out/r16_family_s8.log 2>&1
grep "family sched.: proven" gpurun_out/r16_family_s8.log
(time R3_HARD=1 KAO_DET_CX_STARTS=16 R3_SCHEDS=0 timeout 200 python tools/r3_probe.py family 3.0) > gpurun_out/r16_hard_s16.log 2>&1
grep "family sched.: proven" gpurun_out/r16_hard_s16.log
(time R3_HARD=1 KAO_DET_CX_STARTS=3 R3_SCHEDS=0 timeout 200 python tools/r3_probe.py family 3.0) > gpurun_out/r16_hard_s3.log 2>&1
grep "family sched.: proven" gpurun_out/r16_hard_s3.log
