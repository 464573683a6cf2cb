#!/bin/bash
# round 5, call 40 (final evidence): whole suite, smoke, tools/profile.sh r05_zz (kernel trace + PMC passes of the bench), constants merged, bench with extras
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_zz
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${T}_smoke.log
timeout 1200 bash tools/profile.sh $T 10 > gpurun_out/${T}_profile_stdout.log 2>&1
cp gpurun_out/prof_$T/summary.txt gpurun_out/${T}_final_rocprof_summary.txt
python tools/merge_pmc.py $T | tee gpurun_out/${T}_merge.log
cp profiles/pmc_constants.json gpurun_out/${T}_pmc_constants.json
timeout 1500 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -c 600 gpurun_out/${T}_bench.err
python - <<'P'
import json
b = json.load(open('gpurun_out/r05_zz_bench.json'))
print({k: b[k] for k in ('value', 'ms_per_step', 'time_to_optimal_s')})
print(b['roofline_valu_issue']['frac'], b['roofline_valu_issue']['valu_insts_per_neighbour'], b['roofline']['traffic'])
for t in b['lp_certificate']['topics']: print(t['workload'], t['certificate'], round(t['interior_point_ms']), t['rounded_iterate'])
for t in b['exactness_probe']['topics']: print({k: t[k] for k in t if k in ('brokers', 'partitions', 'status', 'objective', 'certificate', 'seconds', 'seconds_to_proof', 'budget_s')})
for t in b['roofline_big_topic']['topics']: print(t['workload'], t.get('solve_3s'))
P
find gpurun_out/prof_$T -name "*.db" -size +3M -delete; find gpurun_out/prof_$T -name "*.csv" -size +3M -delete
