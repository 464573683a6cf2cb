#!/bin/bash
# round 3, GPU call 25: patience before a new generation on further slack-band topics (P * RF not a multiple of B)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for g in 32 96; do
  echo "KAO_DET_GEN_STALL_L=$g"
  for a in "350 7 2500 1" "450 9 3500 1" "270 6 2200 1" "400 8 3000 2"; do
    KAO_DET_GEN_STALL_L=$g timeout 100 python tools/r3_probe.py solve $a 3,4 3.0
  done
done > gpurun_out/r25_patience_slack.log 2>&1
cat gpurun_out/r25_patience_slack.log
