#!/bin/bash
# round 5, call 11: huge topics in phases: K-search to the first feasible incumbent, KAO-CX (+ K-bound) to its fixpoint, the LP alone, K-search under its prices
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c11
for B in 1 3; do
(time KAO_SOLVE_TRACE=1 timeout 300 python tools/big_topic.py solve drift100k $B) > gpurun_out/${T}_drift100k_${B}s.log 2>&1
grep "workload" gpurun_out/${T}_drift100k_${B}s.log | cut -c1-330
grep "KAO-LP topic 0: [0-9]* iter\|KAO-CX" gpurun_out/${T}_drift100k_${B}s.log | head -12 | cut -c1-200
grep "^\[kao-solve\] launch" gpurun_out/${T}_drift100k_${B}s.log | awk '{print $3":"$5}' | tr '\n' ' ' | cut -c1-600; echo
done
(time timeout 900 python -m pytest tests/test_gpu_lp.py "tests/test_gpu_parity.py::test_solve_is_deterministic" "tests/test_gpu_parity.py::test_drifted_north_star_topic_gets_a_dual_certificate" "tests/test_gpu_parity.py::test_config5_as_one_topic" -q -x -s) > gpurun_out/${T}_pytest.log 2>&1
grep -v "^\[kao" gpurun_out/${T}_pytest.log | tail -8 | cut -c1-300
