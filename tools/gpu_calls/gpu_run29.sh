#!/bin/bash
# round 3, GPU call 29: the whole GPU suite and smoke() on the final state
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 400 python -m pytest tests -m gpu -q) > gpurun_out/r29_pytest.log 2>&1
tail -5 gpurun_out/r29_pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
