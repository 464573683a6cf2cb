#!/bin/bash
# round 3, GPU call 32: kao_solve_capped against the exact joint optimum, per toy (KAO-CX now polishes weighted topics)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 50 python tools/capped_probe.py > gpurun_out/r32_capped.log 2>&1
cat gpurun_out/r32_capped.log | tail -20
