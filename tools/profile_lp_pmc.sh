#!/bin/bash
# rocprofv3 evidence for KAO-LP (run on the GPU box from the repo root): kernel trace + FETCH_SIZE + WRITE_SIZE (separate passes,
# plain launches: KAO_LP_GRAPH=0) of two kao_lp_bound calls on a north-star workload -> per-iteration HBM bytes and kernel time,
# gpurun_out/prof_lp_pmc_<tag>/{summary.txt,constants.json}.
set -u
TAG=${1:-r06}
W=${2:-drift100k}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_lp_pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp KAO_LP_GRAPH=0
cd /tmp
CMD="python $REPO/tools/r6_lp_iter_probe.py $W"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o lp -- $CMD > "$OUT/run_trace.json" 2> "$OUT/trace.err"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o lp -- $CMD > "$OUT/run_fetch.json" 2> "$OUT/fetch.err"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o lp -- $CMD > "$OUT/run_write.json" 2> "$OUT/write.err"
cd "$REPO"
python tools/summarize_lp_pmc.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
