#!/bin/bash
# round 5, call 20 ("final"): whole GPU suite, smoke, bench with extras, tools/profile.sh r05_z (kernel trace + PMC passes of the bench), KAO-LP and a 3-s solve under rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_z
(time timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/${T}_pytest.log 2>&1
tail -6 gpurun_out/${T}_pytest.log | cut -c1-300
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/${T}_smoke.log 2>&1
tail -2 gpurun_out/${T}_smoke.log | cut -c1-300
(time timeout 600 python bench.py) > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err
tail -1 gpurun_out/${T}_bench.log > gpurun_out/${T}_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r05_z_bench.json')); print({k:d[k] for k in ('value','ms_per_step','time_to_optimal_s')}); print(d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('roofline_valu_issue',{}).get('frac'))"
(timeout 900 bash tools/profile.sh r05_z 10) > gpurun_out/${T}_profile.log 2>&1
tail -40 gpurun_out/${T}_profile.log | cut -c1-220
(timeout 200 bash tools/profile_lp.sh r05_z_30k 1000 20 30000) > gpurun_out/${T}_prof_lp.log 2>&1
tail -27 gpurun_out/${T}_prof_lp.log | head -10 | cut -c1-200
(timeout 200 bash tools/profile_solve.sh r05_z_1000x30000 1000 20 30000 3) > gpurun_out/${T}_prof_solve.log 2>&1
tail -18 gpurun_out/${T}_prof_solve.log | cut -c1-200
