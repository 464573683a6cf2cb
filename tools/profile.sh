#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   1. --kernel-trace --stats : per-kernel time (must agree with bench.py's HIP-event average)
#   2. --pmc FETCH_SIZE       : HBM read traffic per dispatch   (separate pass, counters only)
#   3. --pmc WRITE_SIZE       : HBM write traffic per dispatch  (separate pass)
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
set -u
TAG=${1:-r02}
STEPS=${2:-10}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps $STEPS --warmup 2 --no-extras"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/bench_write.json" 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
cd "$REPO"
find "$OUT" -name '*.csv' | head -40
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
