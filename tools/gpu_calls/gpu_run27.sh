#!/bin/bash
# round 3, GPU call 27: k_bound_center (exact line search along the common shifts, once per launch): replay tests on every driver,
# slack-band topics, scale
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dual_bound or bound") > gpurun_out/r27_pytest_bound.log 2>&1
tail -4 gpurun_out/r27_pytest_bound.log
for a in "270 6 2200 1" "350 7 2500 1" "450 9 3500 1" "400 8 3000 1"; do
  R3_SCHEDS=0 timeout 100 python tools/r3_probe.py solve $a 3,4 3.0
done > gpurun_out/r27_slack.log 2>&1
cut -c1-200 gpurun_out/r27_slack.log
(R3_SCHEDS=0 timeout 100 python tools/r3_probe.py scale 3.0) > gpurun_out/r27_scale.log 2>&1
cut -c1-200 gpurun_out/r27_scale.log
