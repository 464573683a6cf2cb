#!/bin/bash
# round 3, GPU call 20: KAO-CX for RF 5..8 and broker weights (whole GPU suite); further starts on the large topics
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 200 python -m pytest tests/test_gpu_cycle.py -m gpu -q -x) > gpurun_out/r20_pytest_cycle.log 2>&1
tail -4 gpurun_out/r20_pytest_cycle.log
(time timeout 600 python -m pytest tests -m gpu -q --deselect tests/test_gpu_cycle.py) > gpurun_out/r20_pytest_rest.log 2>&1
tail -6 gpurun_out/r20_pytest_rest.log
for st in 0 4; do
  for a in "500 10 10000 1" "1000 20 30000 1"; do
    KAO_DET_CX_STARTS=$st timeout 60 python tools/r3_probe.py onetrace $a 3.0 2>/dev/null | tail -1 | cut -c1-100
  done
done > gpurun_out/r20_big_starts.log 2>&1
cat gpurun_out/r20_big_starts.log
