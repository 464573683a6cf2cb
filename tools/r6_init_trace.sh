#!/bin/bash
# round 6: kernel trace of the first launch of config 5 as one topic (K-init against the one-wavefront fill); run from the repo root on the GPU box
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_init; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o init -- python $REPO/tools/r6_init_probe.py > "$OUT/probe.json" 2> "$OUT/trace.err" < /dev/null
cd "$REPO"
timeout 60 python tools/r6_init_trace.py "$OUT"
