#!/bin/bash
# round 4, last call: whole GPU suite on the committed state; kernel trace of one 3-s solve of the drifted 300 x 2000 (second drift seed:
# not proven on solver seed 3) and of 500 x 5000: where the GPU time of a medium solve goes (profiles/r04_zz_solve_*)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_zzz
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/${T}_pytest.log 2>&1
tail -5 gpurun_out/${T}_pytest.log | cut -c1-300
(timeout 200 bash tools/profile_solve.sh r04_zz_500x5000 500 10 5000 3) > gpurun_out/${T}_solve_500.log 2>&1
tail -20 gpurun_out/${T}_solve_500.log | cut -c1-200
