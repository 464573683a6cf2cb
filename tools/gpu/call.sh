#!/bin/bash
# round 5, call 23: min-plus squarings of small matrices split over the midpoints (atomicMin of the composite keys): parity tests, family, kernel trace of a 500 x 5000 solve
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c23
(time timeout 900 python -m pytest tests/test_gpu_cycle.py tests/test_gpu_parity.py -q -x -k "cycle or oracle or bulk or cx or kao_cx or deterministic or fixpoint or working_words") > gpurun_out/${T}_pytest.log 2>&1
tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
(time R3_SCHEDS=0 R3_SEEDS=3,4,5 timeout 900 python tools/r3_probe.py family 3) > gpurun_out/${T}_family.log 2>&1
grep "proven\|real" gpurun_out/${T}_family.log | cut -c1-250
(timeout 200 bash tools/profile_solve.sh r05_c23_500x5000 500 10 5000 3) > gpurun_out/${T}_prof_solve.log 2>&1
head -12 gpurun_out/prof_solve_r05_c23_500x5000/summary.txt | cut -c1-200
