#!/bin/bash
# round 5: the whole drifted family (24 topics x solver seeds 3 / 4 / 5, 3-s limit each) on the final state
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_zz
R3_SCHEDS=0 R3_SEEDS=3,4,5 timeout 600 python tools/r3_probe.py family 3 > gpurun_out/${T}_drift_family.txt 2>&1
grep "proven" gpurun_out/${T}_drift_family.txt | cut -c1-200
python - <<'P'
import re
t = 0.0; n = 0
for l in open('gpurun_out/r05_zz_drift_family.txt'):
    m = re.search(r' ([0-9.]+)s t_best', l)
    if m: t += float(m.group(1)); n += 1
print('solves', n, 'sum of solve seconds', round(t, 2))
P
