#!/bin/bash
# round 6: kernels of KAO-LP by GPU time at 1000 x 100,000 with R racks (arg 1); plain launches (rocprofv3 and the captured iteration do not get on); repo root, GPU box
set -u
R=${1:-40}; REPO=$(pwd); OUT=$REPO/gpurun_out/prof_racks_$R; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp; KAO_LP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o lp -- python $REPO/tools/r6_racks_probe.py $R > "$OUT/probe.txt" 2> "$OUT/err.txt" < /dev/null )
grep "^{" "$OUT/probe.txt"
timeout 60 python - "$OUT" <<'P'
import glob, sqlite3, sys
dbs = glob.glob(sys.argv[1] + "/trace/**/*.db", recursive=True)
if not dbs: print("no database"); sys.exit(0)
c = sqlite3.connect(dbs[0])
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()[:14]:
    short = name.replace("(anonymous namespace)::", "").replace("kao::", "").replace("void ", "").split("(")[0]
    print(f"{short[:50]:50s} calls={calls:6d} total={total/1e6:9.3f} ms avg={avg/1e3:9.1f} us {pct:5.1f}%")
P
