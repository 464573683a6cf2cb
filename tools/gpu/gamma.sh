#!/bin/bash
TAG=${1:-r06_gamma}
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_lp.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/${TAG}_lp_tests.txt
SEEDS="1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16" GAMMAS="0.9995 0" timeout 800 python tools/r6_gamma_probe.py 2>&1 | grep -v "^\[kao" > gpurun_out/${TAG}_gamma.txt
tail -15 gpurun_out/${TAG}_lp_tests.txt; cat gpurun_out/${TAG}_gamma.txt
