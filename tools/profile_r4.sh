#!/bin/bash
# Round-4 profiles (run through gpurun; tools/profile_r4_collect.py turns gpurun_out/r4prof into profiles/r04_*):
#   bench.py and tools/run_configs_r4.py each under (a) --kernel-trace --stats, (b) PMC passes of their own -- never together:
#   FETCH_SIZE / WRITE_SIZE in separate passes (TCC slots), two SQ passes, one cache pass.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- $B --steps 100 --warmup 10 > $O/bench_under_rocprof.json 2> $O/trace.err
BK="$B --steps 20 --warmup 5"
C="python3 $R/tools/run_configs_r4.py"
pass() {   # pass <name> <counters...>: the same counters over the bench line and over the configurations
    n=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" -d $O/k1_$n --output-format csv -- $BK > /dev/null 2> $O/k1_$n.err
    timeout -k 10 400 rocprofv3 --pmc "$@" -d $O/cfg_$n --output-format csv -- $C > $O/cfg_$n.log 2> $O/cfg_$n.err
}
pass fetch FETCH_SIZE TCC_EA0_RDREQ_sum
pass write WRITE_SIZE TCC_EA0_WRREQ_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass misc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAVES
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/cfg_trace --output-format csv -- $C > $O/cfg_run.log 2> $O/cfg_trace.err
timeout -k 10 300 $C > $O/cfg_plain.log 2> $O/cfg_plain.err
ls $O
tail -c 300 $O/bench_under_rocprof.json
