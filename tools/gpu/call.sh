#!/bin/bash
# round 5, call 44: the triangular solves by one workgroup per row tile: LP tests (trace against the restatement), time per iteration, A/B against the single workgroup
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c44
timeout 900 python -m pytest tests/test_gpu_lp.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
for mw in 1 0; do
KAO_LP_TRSV_MW=$mw timeout 600 python - >> gpurun_out/${T}_lp.log 2>&1 <<'P'
import sys, time, os
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
for which in ('drift30k', 'drift100k'):
    t = sy.north_star_topic(which)
    kao.lp_trace(t, max_iters=1)
    b = kao.lp_bound(t)
    r = kao.lp_round(t, salt=1)
    print('KAO_LP_TRSV_MW=' + os.environ['KAO_LP_TRSV_MW'], which, 'certificate', b['bound'], b['iterations'], 'it', round(b['ms'], 1), 'ms =', round(b['ms'] / b['iterations'], 2), 'ms / it | rounded', r['objective'], r['violations'][0], r['iterations'], 'it', round(r['ms_lp']), 'ms', flush=True)
P
done
cat gpurun_out/${T}_lp.log
