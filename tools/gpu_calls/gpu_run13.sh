#!/bin/bash
# round 3, GPU call 14: subproblem candidate depth by prack_hi
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dual_bound_replay") > gpurun_out/r14_pytest_bound.log 2>&1
tail -3 gpurun_out/r14_pytest_bound.log
(timeout 300 python tools/bound_rate.py one multi:512:16 multi:256:16 multi:128:16 multi:256:8) > gpurun_out/r14_bound_rate.log 2>&1
cat gpurun_out/r14_bound_rate.log
cp kafka_assignment_optimizer_amd/libkao_prof.so kafka_assignment_optimizer_amd/libkao.so
(BOUND_RATE_SHAPES=500x10x5000,1000x20x30000 timeout 300 python tools/bound_rate.py multi:512:16 multi:256:16 multi:128:16) > gpurun_out/r14_bound_phases.log 2>&1
grep "it 4500" gpurun_out/r14_bound_phases.log
