cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout -k 10 300 python3 tools/exp_r4_breakdown.py scenes=tm n=7 | grep -E "ao4 |full"
