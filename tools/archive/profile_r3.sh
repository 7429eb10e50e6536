#!/bin/bash
# Round-3 profiles (run through gpurun; summaries are copied into profiles/ afterwards by tools/profile_r3_collect.py):
#   1. bench.py under rocprofv3 --kernel-trace --stats
#   2. PMC passes for K1 of the headline (FETCH_SIZE / WRITE_SIZE in passes of their own, SQ counters)
# never --pmc together with a trace.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- $B --steps 100 --warmup 10 > $O/bench_under_rocprof.json 2> $O/trace.err
BK="python3 $R/bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d $O/k1_fetch --output-format csv -- $BK > /dev/null 2> $O/k1_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum -d $O/k1_write --output-format csv -- $BK > /dev/null 2> $O/k1_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    -d $O/k1_sq --output-format csv -- $BK > /dev/null 2> $O/k1_sq.err
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAVES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum \
    -d $O/k1_misc --output-format csv -- $BK > /dev/null 2> $O/k1_misc.err
C="python3 $R/tools/run_configs.py"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d $O/cfg_fetch --output-format csv -- $C > /dev/null 2> $O/cfg_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum -d $O/cfg_write --output-format csv -- $C > /dev/null 2> $O/cfg_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    -d $O/cfg_sq --output-format csv -- $C > /dev/null 2> $O/cfg_sq.err
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAVES \
    -d $O/cfg_misc --output-format csv -- $C > /dev/null 2> $O/cfg_misc.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/cfg_trace --output-format csv -- $C > $O/cfg_run.log 2> $O/cfg_trace.err
ls $O
tail -c 300 $O/bench_under_rocprof.json
