#!/bin/bash
# round 3, GPU call 33 (the last seconds of the budget): the gated compound-edge layer on the committed 300 x 2000 fixpoint
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 30 python tools/pairs_probe.py 1 > gpurun_out/r33_pairs.log 2>&1
cat gpurun_out/r33_pairs.log | tail -5
