#!/bin/bash
# round 6, final evidence: rocprofv3 passes of the bench (kernel trace + PMC), of the north-star kernels, of KAO-LP (PMC per iteration),
# of one kao_solve of the flagship (plain LP launches), then the whole GPU suite, the smoke test and the bench with extras.
set -u
TAG=${1:-r06_zz}
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 bash tools/profile.sh $TAG 20 > gpurun_out/${TAG}_profile_sh.log 2>&1
timeout 900 bash tools/profile_big.sh $TAG drift30k drift100k cfg5one > gpurun_out/${TAG}_profile_big.log 2>&1
timeout 600 bash tools/profile_lp_pmc.sh $TAG drift100k > gpurun_out/${TAG}_profile_lp_pmc.log 2>&1
KAO_LP_GRAPH=0 timeout 300 bash tools/profile_solve.sh ${TAG}_100k 1000 20 100000 1 > gpurun_out/${TAG}_profile_solve.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1
tail -3 gpurun_out/${TAG}_pytest.txt; tail -2 gpurun_out/${TAG}_smoke.txt
