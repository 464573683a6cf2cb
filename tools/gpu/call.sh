#!/bin/bash
# round 5, last call: whole suite + smoke on the committed state (after the repair-only mode of the host test hook)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_zz
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|ERROR" > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${T}_smoke.log
