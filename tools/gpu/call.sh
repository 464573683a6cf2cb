#!/bin/bash
# round 4, call 16: whole GPU suite with the tightened tolerances; kao_solve_capped (whole prices, raise-only fallback, repair)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r04_c16
(timeout 200 python tools/capped_probe.py capped_toy.json 20; timeout 200 python tools/capped_probe.py capped_medium.json 20) > gpurun_out/${T}_capped.log 2>&1
cat gpurun_out/${T}_capped.log | cut -c1-250
(time timeout 1200 python -m pytest tests -m gpu -q) > gpurun_out/${T}_pytest.log 2>&1
tail -15 gpurun_out/${T}_pytest.log | cut -c1-400
