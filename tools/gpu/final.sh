#!/bin/bash
# the last step of a round's evidence: the cap + 1 topic, the whole suite, smoke and the bench against the committed counter constants
TAG=${1:-r06_zz}
cd "$(dirname "$0")/../.."
LIMIT=1.0 EXPS=default ONLY="cap+1" timeout 300 python tools/r6_sigexp_family.py 2>&1 | grep -v "^\[kao" | head -1
bash tools/gpu/suite.sh $TAG | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -1 gpurun_out/${TAG}_smoke.txt
python bench.py > gpurun_out/${TAG}_bench_stdout.txt 2> gpurun_out/${TAG}_bench_stderr.txt; tail -c 600 gpurun_out/${TAG}_bench_stdout.txt
