#!/bin/bash
# round 5, last call: whole suite on the committed state (centrality correctors on)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_zz
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|ERROR" > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
