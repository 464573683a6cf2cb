#!/bin/bash
# round 4, call 11: large topics, 3-s solves: restarts {256, 1024} x K-bound back-off {on, off}, two solver seeds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_c11
for w in drift30k drift100k; do
  for rs in 256 1024; do
    for rest in 32768 100000000000; do
      for sd in 3 4; do
        R4_RESTARTS=$rs KAO_X_BOUND_REST=$rest timeout 120 python tools/r4_probe.py solve $w 1 3.0 $sd 2>/dev/null | grep '^{' | sed "s/^/rest_beyond $rest seed $sd /" | cut -c1-250
      done
    done
  done
done | tee gpurun_out/${T}_big.log
