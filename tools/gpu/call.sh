#!/bin/bash
# round 5, call 14: the whole GPU suite with k_search_curg restricted to one round of workgroups
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c14
(time timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/${T}_pytest.log 2>&1
tail -8 gpurun_out/${T}_pytest.log | cut -c1-300
