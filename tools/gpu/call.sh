#!/bin/bash
# round 5, final evidence (third take, after the repair phases): whole suite, smoke, bench with extras on the committed state
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_zz
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|ERROR" > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${T}_smoke.log
timeout 1500 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - <<'P'
import json
b = json.load(open('gpurun_out/r05_zz_bench.json'))
print({k: b[k] for k in ('value', 'ms_per_step', 'time_to_optimal_s')})
for t in b['lp_certificate']['topics']: print(t['workload'], t['certificate'], {k: t['rounded_iterate'][k] for k in ('objective', 'violations', 'equals_certificate', 'fractional_partitions')})
for t in b['exactness_probe']['topics']: print({k: t[k] for k in t if k in ('partitions', 'status', 'objective', 'certificate', 'seconds')})
for t in b['roofline_big_topic']['topics']: print(t['workload'], {k: v for k, v in (t.get('solve_3s') or {}).items() if k in ('status', 'objective', 'certificate', 'seconds')})
P
