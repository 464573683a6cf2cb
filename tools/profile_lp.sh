#!/bin/bash
# rocprofv3 kernel trace of KAO-LP alone (kao_lp_bound) on a drifted topic: where an interior-point iteration's GPU time goes.
# Run on the GPU box from the repo root; summary to gpurun_out/prof_lp_<tag>/summary.txt.
set -u
TAG=${1:-r05}
B=${2:-1000}; R=${3:-20}; P=${4:-30000}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_lp_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o lp -- python $REPO/tools/r5_lp_probe.py ${B}x${R}x${P} > "$OUT/lp.txt" 2> "$OUT/trace.err"
cd "$REPO"
python - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import glob, os, sqlite3, sys
out = sys.argv[1]
print(open(os.path.join(out, "lp.txt")).read().strip())
for db in sorted(glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    print("== kernels by total GPU time (rocprofv3 --kernel-trace --stats; the probe runs the solve twice: lp_trace and lp_bound) ==")
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()[:24]:
        short = name.replace("(anonymous namespace)::", "").replace("kao::", "").replace("void ", "").split("(")[0]
        print(f"{short[:60]:60s} calls={calls:6d} total={total/1e6:9.3f} ms avg={avg/1e3:10.2f} us {pct:5.1f}%")
PY
cat "$OUT/summary.txt"
