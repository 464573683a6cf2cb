#!/bin/bash
# round 3, GPU call 11: k_bound_multi rates by slice size / wavefronts; solve trace for the K-bound launch lengths
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 500 python tools/bound_rate.py) > gpurun_out/r11_bound_rate.log 2>&1
cat gpurun_out/r11_bound_rate.log
(R3_SCHEDS=0 KAO_SOLVE_TRACE=1 timeout 300 python tools/r3_probe.py trace 1.0) > gpurun_out/r11_trace.log 2> gpurun_out/r11_trace.err
cat gpurun_out/r11_trace.log
python - <<'PY'
import re
cur=None; rows={}
for l in open('gpurun_out/r11_trace.err'):
    if l.startswith('trace '): cur=l.strip(); rows[cur]=[]; continue
    m=re.search(r'search\+sync ([\d.]+) ms, bound service ([\d.]+) ms \(last K-bound launch ([\d.]+) ms / (\d+) it', l)
    if m and cur: rows[cur].append(tuple(float(x) for x in m.groups()))
for k,v in rows.items():
    v=v[10:]
    if not v: continue
    import statistics as st
    print(k, 'n', len(v), 'search+sync med %.3f'%st.median(x[0] for x in v), 'service med %.3f'%st.median(x[1] for x in v), 'kbound med %.3f ms / %d it'%(st.median(x[2] for x in v), v[-1][3]))
PY
