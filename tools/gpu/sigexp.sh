#!/bin/bash
TAG=${1:-r06_sigexp}
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
HOOK=KAO_LP_SIGEXP SEEDS="${SEEDS:-1 2 3 4 5 6 7 8}" GAMMAS="${EXPS:-3 6 8 12}" timeout 800 python tools/r6_gamma_probe.py 2>&1 | grep -v "^\[kao" > gpurun_out/${TAG}_sigexp.txt
cat gpurun_out/${TAG}_sigexp.txt
