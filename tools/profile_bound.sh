#!/bin/bash
# rocprofv3 evidence for K-bound alone (k_bound_multi / k_bound_center / ...) on large drifted topics, GPU box, repo root:
#   1. --kernel-trace --stats : time per launch (-> microseconds per iteration)
#   2. --pmc SQ_* (own pass)  : instructions, LDS traffic, busy cycles per launch
#   3. --pmc FETCH_SIZE / WRITE_SIZE (own passes): HBM traffic per launch
# usage: profile_bound.sh <tag> ["B R P" ...]      outputs: gpurun_out/prof_<tag>/<BxP>/  (summary.txt by tools/summarize_prof.py)
set -u
TAG=${1:-r04_kbound}
shift
SHAPES=("${@:-1000 20 30000}")
REPO=$(pwd)
export TMPDIR=/tmp
for S in "${SHAPES[@]}"; do
  set -- $S
  OUT=$REPO/gpurun_out/prof_$TAG/$1x$3
  mkdir -p "$OUT"
  cd /tmp
  CMD="python $REPO/tools/bound_only.py $1 $2 $3 1000 4"
  timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $CMD > "$OUT/run_trace.log" 2> "$OUT/trace.err"
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d "$OUT/pmc_sq" -o bench -- $CMD > "$OUT/run_sq.log" 2> "$OUT/sq.err"
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $CMD > "$OUT/run_fetch.log" 2> "$OUT/fetch.err"
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $CMD > "$OUT/run_write.log" 2> "$OUT/write.err"
  cd "$REPO"
  cat "$OUT/run_trace.log"
  python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
  cat "$OUT/summary.txt"
done
