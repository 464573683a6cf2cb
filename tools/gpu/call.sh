#!/bin/bash
# round 4, final state: whole GPU suite, smoke, bench.py with its extras, PMC passes of the bench (profiles/r04_z_*), hard family,
# slack-band topics, large solves
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_z
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/${T}_pytest.log 2>&1
tail -6 gpurun_out/${T}_pytest.log | cut -c1-300
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > gpurun_out/${T}_smoke.log 2>&1
tail -2 gpurun_out/${T}_smoke.log
(time timeout 900 python bench.py) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cut -c1-400 gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err
(time timeout 600 bash tools/profile.sh r04_z 10) > gpurun_out/${T}_profile.log 2>&1
tail -12 gpurun_out/${T}_profile.log | cut -c1-300
(time R3_HARD=1 R3_SEEDS=3,4,5 R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family 3.0) > gpurun_out/${T}_family.log 2>&1
grep -h "proven [0-9]" gpurun_out/${T}_family.log
for w in drift30k drift100k; do for lim in 1.0 3.0; do timeout 60 python tools/r4_probe.py solve $w 1 $lim 3 2>/dev/null | grep '^{' | cut -c1-260; done; done | tee gpurun_out/${T}_big.log
