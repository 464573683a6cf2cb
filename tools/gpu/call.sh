#!/bin/bash
# round 4, final state (after the KAO-CX cadence 8 / 48): whole GPU suite, smoke, bench.py with its extras, hard family, slack topics
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_zz
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/${T}_pytest.log 2>&1
tail -6 gpurun_out/${T}_pytest.log | cut -c1-300
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > gpurun_out/${T}_smoke.log 2>&1
tail -4 gpurun_out/${T}_smoke.log | head -2
(time timeout 900 python bench.py) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cut -c1-300 gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err
(time R3_HARD=1 R3_SEEDS=3,4,5 R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family 3.0) > gpurun_out/${T}_family.log 2>&1
grep -h "proven [0-9]" gpurun_out/${T}_family.log
for shape in "270 6 2200" "350 7 2500" "450 9 3500"; do
  timeout 100 python tools/r3_probe.py solve $shape 1 3,4,5 3.0 2>&1 | grep "solve seed" | cut -c1-120
done | tee gpurun_out/${T}_slack.log
