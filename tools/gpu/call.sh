#!/bin/bash
# round 4, call 2: the whole GPU suite with the TEAM kernel (k_team) as the default for topics in global memory (new replay test
# against the port), then launch time / depth against the team size on the north-star topics and whole solves with and without.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
(time timeout 600 python -m pytest tests -m gpu -x -q) > gpurun_out/r04_c2_pytest.log 2>&1
tail -15 gpurun_out/r04_c2_pytest.log
(time timeout 200 python tools/r4_probe.py team drift30k 256 1,2,4,8 6) > gpurun_out/r04_c2_team_30k.log 2>&1
(time timeout 200 python tools/r4_probe.py team drift100k 256 1,8 4) > gpurun_out/r04_c2_team_100k.log 2>&1
(time timeout 200 python tools/r4_probe.py team drift30k 64,128,512 8 6) >> gpurun_out/r04_c2_team_30k.log 2>&1
cat gpurun_out/r04_c2_team_30k.log gpurun_out/r04_c2_team_100k.log | cut -c1-420
(time timeout 120 python tools/r4_probe.py solve drift30k 1,8,4 3.0) > gpurun_out/r04_c2_solve_30k.log 2>&1
(time timeout 120 python tools/r4_probe.py solve drift100k 1,8 3.0) > gpurun_out/r04_c2_solve_100k.log 2>&1
cat gpurun_out/r04_c2_solve_30k.log gpurun_out/r04_c2_solve_100k.log | cut -c1-700
