#!/bin/bash
# whole suite, fuzz against HiGHS, the seed sweep: after a change of the interior-point rule
TAG=${1:-r06_robust}
cd "$(dirname "$0")/../.."
bash tools/gpu/suite.sh $TAG | tail -4
timeout 900 python tools/r6_fuzz_solve.py 1500 600 30000 2>&1 | grep -v "^\[kao" | tail -4 > gpurun_out/${TAG}_fuzz.txt; cat gpurun_out/${TAG}_fuzz.txt
timeout 600 python tools/r6_fuzz_mid.py 2>&1 | grep -v "^\[kao" | tail -4 > gpurun_out/${TAG}_fuzz_mid.txt; cat gpurun_out/${TAG}_fuzz_mid.txt
timeout 600 python tools/r6_seed_sweep.py 2>&1 | grep -v "^\[kao" > gpurun_out/${TAG}_seed_sweep.txt; cat gpurun_out/${TAG}_seed_sweep.txt
