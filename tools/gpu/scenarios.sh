#!/bin/bash
# the operational scenarios, the multi-topic calls, the model variants and the edges (tools/r6_scenarios*.py) in one call
TAG=${1:-r06_scen}
cd "$(dirname "$0")/../.."
for n in "" 2 3 4; do
  echo "== tools/r6_scenarios$n.py =="
  timeout 400 python tools/r6_scenarios$n.py 2>&1 | grep -v "^\[kao\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
done > gpurun_out/${TAG}_scenarios.txt 2>&1
cat gpurun_out/${TAG}_scenarios.txt
