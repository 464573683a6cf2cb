#!/bin/bash
# round 3, GPU call 6: rocprofv3 evidence -- the bench kernel at HEAD, the north-star regime (1000 x 30000 drifted, config 5 as one topic)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 700 bash tools/profile.sh r03_b 10) > gpurun_out/r6_profile.log 2>&1
(time timeout 900 bash tools/profile_big.sh r03 drift30k cfg5one) > gpurun_out/r6_profile_big.log 2>&1
tail -12 gpurun_out/r6_profile.log | cut -c1-400; tail -40 gpurun_out/r6_profile_big.log | cut -c1-300
