#!/bin/bash
# round 3, GPU call 28 (final state with k_bound_center): whole GPU suite, family (seed 3) + hard half (seeds 4, 5), bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 700 python -m pytest tests -m gpu -q) > gpurun_out/r28_pytest.log 2>&1
tail -4 gpurun_out/r28_pytest.log
(time R3_SCHEDS=0 timeout 300 python tools/r3_probe.py family 3.0) > gpurun_out/r28_family.log 2>&1
grep "proven" gpurun_out/r28_family.log
(time R3_HARD=1 R3_SEEDS=4,5 R3_SCHEDS=0 timeout 300 python tools/r3_probe.py family 3.0) > gpurun_out/r28_hard.log 2>&1
grep "proven" gpurun_out/r28_hard.log
(time timeout 400 python bench.py) > gpurun_out/r28_bench.json 2> gpurun_out/r28_bench.err
tail -c 300 gpurun_out/r28_bench.json
