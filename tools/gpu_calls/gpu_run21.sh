#!/bin/bash
# round 3, GPU call 21: per-launch timing trace (K-search launch vs K-bound launch) for re-fitting the deterministic schedule's table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
(R3_SCHEDS=0 timeout 120 python tools/r3_probe.py trace 1.0) > gpurun_out/r21_trace.out 2> gpurun_out/r21_trace.err
grep -c launch gpurun_out/r21_trace.err
