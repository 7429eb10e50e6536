cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_chain_counts.py -x -q -m gpu -k twins 2>&1 | grep -E "DIAG|passed|failed" | head
timeout -k 10 600 python tools/exp_r4_ao_util.py 2>&1 | grep UTIL
