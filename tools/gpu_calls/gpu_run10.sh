#!/bin/bash
# round 3, GPU call 10: k_bound_multi (persistent multi-workgroup K-bound): replay tests first, then rates, then the rest
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dual_bound") > gpurun_out/r10_pytest_bound.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r10_pytest_bound.log
tail -8 gpurun_out/r10_pytest_bound.log
(timeout 400 python tools/bound_rate.py) > gpurun_out/r10_bound_rate.log 2>&1
cat gpurun_out/r10_bound_rate.log
(time timeout 1200 python -m pytest tests -m gpu -q -k "not dual_bound") > gpurun_out/r10_pytest_rest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r10_pytest_rest.log
tail -8 gpurun_out/r10_pytest_rest.log
(time R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r10_family.log 2>&1
grep "family sched.: proven" gpurun_out/r10_family.log; grep "scale" gpurun_out/r10_family.log | cut -c1-190
