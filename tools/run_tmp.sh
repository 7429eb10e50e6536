cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_tiletags.py tests/test_gpu_temporal.py tests/test_gpu_shard.py -q -m gpu -x 2>&1 | tail -4
timeout -k 10 300 python3 tools/exp_r4_inflight.py
for V in w6 w5 w4; do
  VRT_LIB=$GRAFT_REPO_ROOT/voxel-raytracing_amd/csrc/libvrt_hip_$V.so timeout -k 10 300 python3 tools/exp_r4_breakdown.py scenes=m n=5 | grep full | sed "s/^/$V /"
done
timeout -k 10 300 python3 tools/exp_r4_breakdown.py scenes=m n=5 | grep full | sed "s/^/w7 /"
