#!/bin/bash
# round 5: rocprofv3 kernel trace of ONE kao_solve of the drifted north-star topic (plain LP launches: the tool crashes in the graph capture at this size)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_zz
KAO_LP_GRAPH=0 timeout 600 bash tools/profile_solve.sh ${T}_100k 1000 20 100000 3 > /dev/null 2>&1
cp gpurun_out/prof_solve_${T}_100k/summary.txt gpurun_out/${T}_solve_1000x100000_rocprof_summary.txt; head -22 gpurun_out/${T}_solve_1000x100000_rocprof_summary.txt | cut -c1-200
find gpurun_out/prof_solve_${T}_100k -name "*.db" -size +5M -delete; find gpurun_out/prof_solve_${T}_100k -name "*.csv" -size +5M -delete
