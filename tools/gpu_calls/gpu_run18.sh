#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for a in "400 8 3000 1" "300 6 2000 2" "250 10 2000 1"; do
  timeout 60 python tools/r3_probe.py onetrace $a 3.0 2>&1 | grep -E "KAO-CX|generation [0-9]+, its|OPTIMAL|TIME" | awk '!/generation/ || !seen[$0]++' > gpurun_out/r18_trace_$(echo $a | tr ' ' '_').log
done
