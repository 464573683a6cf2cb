#!/bin/bash
# round 4, call 7: REPLACE scan over the tournament's two best slots (port restated): GPU suite, bench value, family, slack topics
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/r04_c7_pytest.log 2>&1
tail -12 gpurun_out/r04_c7_pytest.log | cut -c1-300
(time timeout 600 python bench.py --no-extras) > gpurun_out/r04_c7_bench.json 2> gpurun_out/r04_c7_bench.err
cut -c1-1500 gpurun_out/r04_c7_bench.json
(time R3_HARD=1 R3_SEEDS=3,4,5 R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family 3.0) > gpurun_out/r04_c7_family.log 2>&1
grep -h "proven" gpurun_out/r04_c7_family.log
for shape in "270 6 2200" "350 7 2500" "450 9 3500" "400 8 3000"; do
  timeout 100 python tools/r3_probe.py solve $shape 1 3,4,5 3.0 2>&1 | grep "solve seed"
done > gpurun_out/r04_c7_slack.log
cat gpurun_out/r04_c7_slack.log | cut -c1-120
