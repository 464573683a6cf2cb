#!/bin/bash
# round 5, call 16: kao_solve_capped as a portfolio over the price granularity (4, 2, 1); the capped tests; the tests touched by k_search_curg
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c16
(timeout 300 python tools/capped_probe.py capped_medium.json 10; timeout 300 python tools/capped_probe.py capped_medium.json 1; timeout 300 python tools/capped_probe.py capped_toy.json 10) > gpurun_out/${T}_capped.log 2>&1
cat gpurun_out/${T}_capped.log | cut -c1-250
(time timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "capped or global_memory or team_search or working_words or further_kao") > gpurun_out/${T}_pytest.log 2>&1
tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
