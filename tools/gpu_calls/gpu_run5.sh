#!/bin/bash
# round 3, GPU call 5: wide K-search (paired tournament slots, two scan rounds per trip), K-bound enqueued behind K-search, CX cadence by size
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/r5_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_pytest.log
(time timeout 300 python bench.py --no-extras) > gpurun_out/r5_bench_quick.json 2> gpurun_out/r5_bench_quick.err
(time R3_SCHEDS=0 timeout 600 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r5_family.log 2>&1
(time R3_SCHEDS=0 timeout 300 python tools/r3_probe.py trace) > gpurun_out/r5_trace.log 2> gpurun_out/r5_trace.err
tail -6 gpurun_out/r5_pytest.log; grep "family sched.: proven" gpurun_out/r5_family.log; grep "scale" gpurun_out/r5_family.log | cut -c1-150; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_bench_quick.json
