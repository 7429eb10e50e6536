cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_chain_counts.py tests/test_gpu_bricks.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -6
for pk in 0 1; do
  timeout -k 10 300 python3 tools/exp_r4_breakdown.py scenes=b n=5 packed_bounces=$pk | sed "s/^/packed=$pk /"
done
