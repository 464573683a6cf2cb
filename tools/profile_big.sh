#!/bin/bash
# rocprofv3 evidence for the north-star regime (run on the GPU box from the repo root): per workload
#   steps: kernel trace + FETCH_SIZE + WRITE_SIZE + SQ counters of equal K-search / K-eval launches (separate passes)
#   solve: kernel trace of one whole kao_solve (K-search, K-eval, K-bound, KAO-CX: shares of the GPU time)
# Every pass runs under its own timeout (a 9-counter SQ pass once hung for 15 minutes on the 30,000-partition topic).
# Summaries and per-kernel roofline lines: gpurun_out/prof_big_<tag>/<workload>/summary.txt (+ constants.json).
set -u
TAG=${1:-r03}
shift
WORKLOADS=${*:-drift30k cfg5one}
REPO=$(pwd)
export TMPDIR=/tmp
for W in $WORKLOADS; do
  OUT=$REPO/gpurun_out/prof_big_$TAG/$W
  mkdir -p "$OUT"
  cd /tmp
  STEPS="python $REPO/tools/big_topic.py steps $W 6"
  timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o steps -- $STEPS > "$OUT/steps_trace.json" 2> "$OUT/trace.err"
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o steps -- $STEPS > "$OUT/steps_fetch.json" 2> "$OUT/fetch.err"
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o steps -- $STEPS > "$OUT/steps_write.json" 2> "$OUT/write.err"
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d "$OUT/pmc_sq" -o steps -- $STEPS > "$OUT/steps_sq.json" 2> "$OUT/sq.err"
  timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/solve_trace" -o solve -- python $REPO/tools/big_topic.py solve $W 3 > "$OUT/solve.json" 2> "$OUT/solve.err"
  cd "$REPO"
  python tools/summarize_big.py "$OUT" > "$OUT/summary.txt" 2>&1
  cat "$OUT/summary.txt"
done
