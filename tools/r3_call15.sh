cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/debug_flake.py 2>&1 | tail -12
