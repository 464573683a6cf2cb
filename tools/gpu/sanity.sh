cd "$GRAFT_REPO_ROOT" || exit 1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 120 python tools/capped_probe.py capped_toy.json 5 2>&1 | tail -3
