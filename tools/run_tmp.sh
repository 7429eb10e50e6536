cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bricks.py tests/test_gpu_chain_counts.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -4
for ab in 1 0 1 0; do
timeout -k 10 300 python3 tools/exp_r4_breakdown.py scenes=b n=5 ao_batch=$ab | grep -E "ao4|full" | sed "s/^/ao_batch=$ab /"
done
