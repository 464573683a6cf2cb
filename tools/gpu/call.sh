#!/bin/bash
# round 5, call 1: KAO-LP on the device for the first time -- trace against the scalar restatement, certificates, times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c01
(time timeout 900 python tools/r5_lp_probe.py 100x5x1000 130x5x1000 270x6x2200 450x9x3500 500x10x5000 1000x20x30000) > gpurun_out/${T}_lp_probe.log 2>&1
tail -60 gpurun_out/${T}_lp_probe.log | cut -c1-250
