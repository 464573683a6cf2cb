#!/bin/bash
# rocprofv3 kernel trace of ONE kao_solve call on a large drifted topic: where the GPU time of K-search, K-eval, K-bound and the
# KAO-CX kernels goes (run on the GPU box from the repo root; summary to gpurun_out/prof_solve_<tag>/summary.txt).
set -u
TAG=${1:-r02}
B=${2:-1000}; R=${3:-20}; P=${4:-30000}; BUDGET=${5:-3}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_solve_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o solve -- python $REPO/tools/r3_probe.py solve $B $R $P 1 3 $BUDGET > "$OUT/solve.txt" 2> "$OUT/trace.err"
cd "$REPO"
python - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import glob, os, sqlite3, sys
out = sys.argv[1]
print(open(os.path.join(out, "solve.txt")).read().strip())
for db in sorted(glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    print("== kernels by total GPU time (rocprofv3 --kernel-trace --stats) ==")
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()[:16]:
        short = name.replace("(anonymous namespace)::", "").replace("kao::", "").replace("void ", "").split("(")[0]
        print(f"{short[:60]:60s} calls={calls:6d} total={total/1e6:9.3f} s avg={avg/1e3:10.2f} ms {pct:5.1f}%")
PY
cat "$OUT/summary.txt"
