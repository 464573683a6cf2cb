#!/bin/bash
# rocprofv3 kernel trace of kao_solve on the wide golden family (K-search + K-eval + K-bound), GPU box, repo root.
set -u
TAG=${1:-r01_bound}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o wide -- python $REPO/tools/wide_family_solve.py > "$OUT/wide.log" 2> "$OUT/trace.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d "$OUT/pmc_sq" -o wide -- python $REPO/tools/wide_family_solve.py > "$OUT/wide_sq.log" 2> "$OUT/sq.err"
cd "$REPO"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
