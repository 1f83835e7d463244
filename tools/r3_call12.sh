cd $GRAFT_REPO_ROOT/tools/bin/r02tree
for rep in 1 2 3 4 5 6; do timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done
