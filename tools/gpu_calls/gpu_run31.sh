#!/bin/bash
# round 3, GPU call 31: rocprofv3 kernel trace of one whole kao_solve in the final state, drifted 1000 x 30000 topic (3 s)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_big_r03f/drift30k; mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -d "$OUT/solve_trace" -o solve -- python $REPO/tools/big_topic.py solve drift30k 3 > "$OUT/solve.json" 2> "$OUT/solve.err")
python tools/summarize_big.py "$OUT" > "$OUT/summary.txt" 2>&1
tail -22 "$OUT/summary.txt"
