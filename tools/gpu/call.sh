#!/bin/bash
# round 4, call 26: KAO-CX cadence 4 / 24 and 6 / 36 on the hard half of the family (adopted: 8 / 48), and on the 500-broker scale rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_c26
for cad in "4 24" "6 36"; do
  set -- $cad
  (KAO_DET_CX_STALL_L=$1 KAO_DET_CX_DUE_L=$2 R3_HARD=1 R3_SEEDS=3,4,5 R3_SCHEDS=0 timeout 400 python tools/r3_probe.py family 3.0) > gpurun_out/${T}_family_cx_$1_$2.log 2>&1
  grep -h "proven [0-9]" gpurun_out/${T}_family_cx_$1_$2.log | sed "s/^/cadence $1 $2: /"
  for shape in "500 10 5000" "500 10 10000"; do
    KAO_DET_CX_STALL_L=$1 KAO_DET_CX_DUE_L=$2 timeout 100 python tools/r3_probe.py solve $shape 1 3,4 3.0 2>&1 | grep "solve seed" | cut -c1-110 | sed "s/^/cadence $1 $2: /"
  done
done | tee gpurun_out/${T}_summary.log
for shape in "500 10 5000" "500 10 10000"; do
  timeout 100 python tools/r3_probe.py solve $shape 1 3,4 3.0 2>&1 | grep "solve seed" | cut -c1-110 | sed "s/^/default: /"
done | tee -a gpurun_out/${T}_summary.log
