cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -q 2>&1 | tail -8
