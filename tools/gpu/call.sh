#!/bin/bash
# round 4, call 22 (row loads batched behind an acquire fence): k_bound_multi publishes a slice's counts as a row (plain stores) instead of B 64-bit atomics: replay tests on every
# driver, microseconds per iteration (tools/bound_rate.py), a 3-s solve of 1000 x 30000
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_c22
(time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dual or bound or certificate or slack or deterministic") > gpurun_out/${T}_pytest.log 2>&1
tail -4 gpurun_out/${T}_pytest.log | cut -c1-300
(time BOUND_RATE_SHAPES=300x6x2000,500x10x5000,500x10x10000,1000x20x30000 timeout 300 python tools/bound_rate.py multi:512:16 multi:1024:16) > gpurun_out/${T}_bound_rate.log 2>&1
cat gpurun_out/${T}_bound_rate.log | cut -c1-300
for sd in 3 4; do timeout 60 python tools/r4_probe.py solve drift30k 1 3.0 $sd 2>/dev/null | grep '^{' | cut -c1-420; done | tee gpurun_out/${T}_big.log
