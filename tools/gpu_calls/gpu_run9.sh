#!/bin/bash
# round 3, GPU call 9: K-bound for RF 5..8 and broker weights (replay + solves), whole GPU suite, microbench with the real clock
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/r9_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r9_pytest.log
(cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate valu_rate.hip 2>/dev/null && timeout 120 /tmp/valu_rate) > gpurun_out/r9_valu_rate.log 2>&1
tail -15 gpurun_out/r9_pytest.log; cat gpurun_out/r9_valu_rate.log
