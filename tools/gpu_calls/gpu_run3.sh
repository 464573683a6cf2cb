#!/bin/bash
# round 3, GPU call 3: u16 band state, restored KAO-CX cadence, multi-seed probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/r3_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_pytest.log
(time timeout 300 python bench.py --no-extras) > gpurun_out/r3_bench_quick.json 2> gpurun_out/r3_bench_quick.err
(time R3_SCHEDS=0 timeout 600 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r3_family.log 2>&1
(time timeout 600 python tools/r3_probe.py seeds 3.0) > gpurun_out/r3_seeds.log 2>&1
tail -8 gpurun_out/r3_pytest.log; grep "family sched.: proven" gpurun_out/r3_family.log; tail -1 gpurun_out/r3_seeds.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r3_bench_quick.json
