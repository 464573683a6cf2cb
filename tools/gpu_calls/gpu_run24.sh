#!/bin/bash
# round 3, GPU call 24: patience before a new generation on the drifted 400 x 3000 topic
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for g in 32 96 100000; do
  echo "KAO_DET_GEN_STALL_L=$g"
  KAO_DET_GEN_STALL_L=$g timeout 100 python tools/r3_probe.py solve 400 8 3000 1 3,4,5 3.0
done > gpurun_out/r24_patience.log 2>&1
cat gpurun_out/r24_patience.log
