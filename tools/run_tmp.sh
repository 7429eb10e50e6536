cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu -x 2>&1 | tail -12
