#!/bin/bash
# round 4, call 5: KAO-CX with one slack node per rack (+ a global one): GPU suite (matrices / seeds / rounds against the restated
# oracle), the slack-band topics of round 3 (270x2200, 350x2500, 450x3500, 400x3000), the hard half of the drifted family
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/r04_c5_pytest.log 2>&1
tail -5 gpurun_out/r04_c5_pytest.log
for shape in "270 6 2200" "350 7 2500" "450 9 3500" "400 8 3000"; do
  timeout 100 python tools/r3_probe.py solve $shape 1 3,4,5 3.0 2>&1 | grep "solve seed"
done > gpurun_out/r04_c5_slack.log
cat gpurun_out/r04_c5_slack.log
(time R3_HARD=1 R3_SEEDS=3,4,5 R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family 3.0) > gpurun_out/r04_c5_family.log 2>&1
grep -h "proven" gpurun_out/r04_c5_family.log
