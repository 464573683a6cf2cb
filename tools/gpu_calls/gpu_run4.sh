#!/bin/bash
# round 3, GPU call 4: population generations in the deterministic schedule
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/r4_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_pytest.log
(time R3_SCHEDS=0 timeout 600 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r4_family.log 2>&1
(time timeout 600 python tools/r3_probe.py goldens) > gpurun_out/r4_goldens.log 2>&1
tail -8 gpurun_out/r4_pytest.log; grep "family sched.: proven" gpurun_out/r4_family.log; cat gpurun_out/r4_goldens.log | cut -c1-150
