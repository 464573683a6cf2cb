#!/bin/bash
# round 3, GPU call 8: K-bound launch sequence as a hipGraph (A/B), tightened tests, re-fitted launch lengths
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/r8_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r8_pytest.log
(KAO_BOUND_GRAPH=1 timeout 300 python tools/bound_rate.py 512; KAO_BOUND_GRAPH=0 timeout 300 python tools/bound_rate.py 512) > gpurun_out/r8_bound_rate.log 2>&1
(time R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r8_family.log 2>&1
(time R3_SCHEDS=0 KAO_BOUND_GRAPH=0 timeout 200 python tools/r3_probe.py scale 3.0) > gpurun_out/r8_scale_nograph.log 2>&1
tail -6 gpurun_out/r8_pytest.log; cat gpurun_out/r8_bound_rate.log; grep "family sched.: proven" gpurun_out/r8_family.log; grep "scale" gpurun_out/r8_family.log gpurun_out/r8_scale_nograph.log | cut -c1-190
