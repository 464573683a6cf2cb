#!/bin/bash
# second SQ counter pass: where do wave cycles go (issue vs waits)?
set -u
TAG=${1:-r01}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
BENCH="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --eval-bench 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d "$OUT/pmc_sq2" -o bench -- $BENCH > /dev/null 2> "$OUT/sq2.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq3" -o bench -- $BENCH > /dev/null 2> "$OUT/sq3.err"
cd "$REPO"
python - "$OUT" <<'PY'
import sqlite3, glob, sys, os
from collections import defaultdict
out=sys.argv[1]
for sub in ("pmc_sq2","pmc_sq3"):
    for db in glob.glob(os.path.join(out,sub,"**","*.db"),recursive=True):
        c=sqlite3.connect(db); g=defaultdict(lambda: defaultdict(list))
        for name,gs,wg,cn,val in c.execute("select kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection"):
            if "k_search" in name: g[gs//wg][cn].append(val)
        for blocks,d in sorted(g.items()):
            print(sub,"k_search workgroups",blocks,{k: "%.4g"%(sum(v)/len(v)) for k,v in sorted(d.items())})
PY
