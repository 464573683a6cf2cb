#!/bin/bash
# NOT RUN YET (round 3 ended with 0.2 GPU-minutes): the first measurement of the next round -- the compound-edge layer of KAO-CX
# (KAO_CX_PAIRS=1, elite descents only) against the default on the hard half of the drifted family, three solver seeds, and on
# the whole GPU suite (test_compound_edge_layer_lifts_the_committed_fixpoint runs it on one fixpoint).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for on in 0 1; do
  (time KAO_CX_PAIRS=$on R3_HARD=1 R3_SEEDS=3,4,5 R3_SCHEDS=0 timeout 300 python tools/r3_probe.py family 3.0) > gpurun_out/next_pairs_$on.log 2>&1
  echo "KAO_CX_PAIRS=$on"; grep "proven" gpurun_out/next_pairs_$on.log
done
KAO_CX_TRACE=1 KAO_CX_PAIRS=1 timeout 60 python tools/r3_probe.py solve 300 6 2000 2 3 3.0 2>&1 | grep -E "pairs:|solve seed" | tail -12
