cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_tiletags.py tests/test_gpu_shard.py -q -m gpu -x 2>&1 | tail -4
timeout -k 10 600 python3 tools/exp_shardstep.py only inplace 2>&1 | grep -v "^$" | tail -12
