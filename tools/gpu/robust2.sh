#!/bin/bash
# whole suite, fuzz against HiGHS, the north star's relatives: after a change of the interior-point rule
TAG=${1:-r06_robust}
cd "$(dirname "$0")/../.."
bash tools/gpu/suite.sh $TAG | tail -12
timeout 900 python tools/r6_fuzz_solve.py 1500 600 30000 2>&1 | grep -v "^\[kao" | tail -4 > gpurun_out/${TAG}_fuzz.txt; cat gpurun_out/${TAG}_fuzz.txt
LIMIT=1.0 EXPS=${EXPS:-10} timeout 800 python tools/r6_sigexp_family.py 2>&1 | grep -v "^\[kao" > gpurun_out/${TAG}_family.txt; cat gpurun_out/${TAG}_family.txt
