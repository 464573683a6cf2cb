#!/bin/bash
# round 3, GPU call 12: K-bound drivers after the pools rewrite; phase profile of k_bound_multi (libkao_prof.so, -DKAO_BOUND_PROFILE)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python tools/bound_rate.py one step:512 multi:512:8 multi:512:16) > gpurun_out/r12_bound_rate.log 2>&1
cat gpurun_out/r12_bound_rate.log
cp kafka_assignment_optimizer_amd/libkao_prof.so kafka_assignment_optimizer_amd/libkao.so
(BOUND_RATE_SHAPES=300x6x2000,500x10x5000,1000x20x30000 timeout 300 python tools/bound_rate.py multi:512:8 multi:512:16 multi:1024:16) > gpurun_out/r12_bound_phases.log 2>&1
cat gpurun_out/r12_bound_phases.log
