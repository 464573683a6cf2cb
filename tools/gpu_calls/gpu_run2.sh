#!/bin/bash
# round 3, GPU call 2: band-state K-search (bit-exact replay tests), new deterministic-schedule constants, profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/r2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest.log
(time timeout 300 python bench.py --no-extras) > gpurun_out/r2_bench_quick.json 2> gpurun_out/r2_bench_quick.err
(time R3_SCHEDS=0 timeout 600 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r2_family.log 2>&1
(time timeout 600 bash tools/profile.sh r03_a 10) > gpurun_out/r2_profile.log 2>&1
tail -15 gpurun_out/r2_pytest.log; grep "family sched.: proven" gpurun_out/r2_family.log; tail -3 gpurun_out/r2_bench_quick.json | cut -c1-600
