#!/bin/bash
# the whole GPU suite, its tail kept under gpurun_out/<tag>_pytest.txt
TAG=${1:-r06_suite}
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > gpurun_out/${TAG}_pytest.txt
tail -40 gpurun_out/${TAG}_pytest.txt
