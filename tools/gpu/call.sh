#!/bin/bash
# round 5, call 25: two restarts per compute unit on HBM topics as kao_solve's default: scale rows, drifted north-star topic, the tests that touch it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c25
(SEEDS=1,2,3,4,5 BUDGET=4 timeout 600 python tools/r5_solve_probe.py 1000x20x30000 500x10x10000) > gpurun_out/${T}_solve.log 2>&1
cut -c1-215 gpurun_out/${T}_solve.log
for B in 1 3; do (timeout 300 python tools/big_topic.py solve drift100k $B) 2>&1 | grep workload | cut -c1-330; done
(time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "north_star or config5 or deterministic or large_topic or global_memory") > gpurun_out/${T}_pytest.log 2>&1
tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
