cd $GRAFT_REPO_ROOT
for rep in 1 2; do timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3; done
