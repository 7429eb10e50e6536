set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for N in 1 8; do
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/cmp_sq_$N --output-format csv -- python3 $R/tools/exp_shardone.py $N > $R/gpurun_out/cmp_$N.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $R/gpurun_out/cmp_mem_$N --output-format csv -- python3 $R/tools/exp_shardone.py $N > /dev/null 2>&1
done
tail -1 $R/gpurun_out/cmp_1.log $R/gpurun_out/cmp_8.log
