#!/bin/bash
# determinism and a fresh fuzz range on the final library
TAG=${1:-r06_last}
cd "$(dirname "$0")/../.."
{ echo "== tools/r6_determinism.py =="; timeout 600 python tools/r6_determinism.py 2>&1 | grep -v "^\[kao\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids";
  echo "== tools/r6_fuzz_solve.py 1500 600 50000 =="; timeout 900 python tools/r6_fuzz_solve.py 1500 600 50000 2>&1 | grep -v "^\[kao\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -4; } > gpurun_out/${TAG}_checks.txt 2>&1
cat gpurun_out/${TAG}_checks.txt
