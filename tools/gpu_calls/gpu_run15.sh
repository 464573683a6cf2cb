#!/bin/bash
# round 3, GPU call 15: packed count atomics + ranked phase R; replay tests, rates, family/scale probe, phase profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dual_bound") > gpurun_out/r15_pytest_bound.log 2>&1
tail -3 gpurun_out/r15_pytest_bound.log
(timeout 300 python tools/bound_rate.py one multi:512:16 multi:512:8 multi:1024:16 multi:256:16) > gpurun_out/r15_bound_rate.log 2>&1
cat gpurun_out/r15_bound_rate.log
(time R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r15_family.log 2>&1
grep "family sched.: proven" gpurun_out/r15_family.log; grep "scale" gpurun_out/r15_family.log | cut -c1-190
cp kafka_assignment_optimizer_amd/libkao_prof.so kafka_assignment_optimizer_amd/libkao.so
(BOUND_RATE_SHAPES=500x10x5000,1000x20x30000 timeout 300 python tools/bound_rate.py multi:512:16 multi:1024:16) > gpurun_out/r15_bound_phases.log 2>&1
grep "it 4500" gpurun_out/r15_bound_phases.log
