cd $GRAFT_REPO_ROOT
python tools/fixed_cost_probe.py 2>&1 | tail -12
