#!/bin/bash
# round 3, GPU call 1: full GPU suite on the new code, probes for the deterministic schedule, bench baseline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/r1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r1_pytest.log
(time timeout 200 python tools/r3_probe.py trace) > gpurun_out/r1_trace.log 2> gpurun_out/r1_trace.err
(time timeout 400 python tools/r3_probe.py determinism) > gpurun_out/r1_determinism.log 2>&1
(time timeout 600 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r1_family.log 2>&1
(time timeout 400 python tools/r3_probe.py dump 3.0) > gpurun_out/r1_dump.log 2>&1
(time timeout 600 python bench.py) > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err
tail -5 gpurun_out/r1_pytest.log; tail -3 gpurun_out/r1_determinism.log; grep "family sched.: proven" gpurun_out/r1_family.log
