#!/bin/bash
# round 3, GPU call 7: cooperative K-eval, consecutive tournament partitions; north-star steps and profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/r7_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r7_pytest.log
(time timeout 200 python bench.py --no-extras) > gpurun_out/r7_bench_quick.json 2> gpurun_out/r7_bench_quick.err
for W in drift5k drift30k cfg5one; do timeout 200 python tools/big_topic.py steps $W 6; done > gpurun_out/r7_big_steps.log 2>&1
(time R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r7_family.log 2>&1
(time KAO_WIDE_DUAL_ONLY=1 timeout 200 python tools/wide_family_solve.py) > gpurun_out/r7_wide.log 2>&1
(time timeout 900 bash tools/profile_big.sh r03b drift30k) > gpurun_out/r7_profile_big.log 2>&1
tail -6 gpurun_out/r7_pytest.log; grep "family sched.: proven" gpurun_out/r7_family.log; grep "scale" gpurun_out/r7_family.log | cut -c1-150; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r7_bench_quick.json; cat gpurun_out/r7_big_steps.log | cut -c1-420; tail -3 gpurun_out/r7_wide.log | cut -c1-400; tail -25 gpurun_out/r7_profile_big.log | cut -c1-330
