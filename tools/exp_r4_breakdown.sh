#!/bin/bash
# tools/exp_r4_breakdown.py plain and under two --pmc passes (never together with a trace)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_breakdown${TAG:+_$TAG}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/tools/exp_r4_breakdown.py "$@" > $O/plain.log 2> $O/plain.err
cat $O/plain.log
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM \
    -d $O/sq --output-format csv -- python3 $R/tools/exp_r4_breakdown.py "$@" > $O/sq.log 2> $O/sq.err
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc --output-format csv -- python3 $R/tools/exp_r4_breakdown.py "$@" > $O/tcc.log 2> $O/tcc.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/wr --output-format csv -- python3 $R/tools/exp_r4_breakdown.py "$@" > $O/wr.log 2> $O/wr.err
python3 $R/tools/exp_r4_breakdown_pmc.py $O/sq.log $O/sq/*/*_counter_collection.csv | tee $O/sq_summary.txt
python3 $R/tools/exp_r4_breakdown_pmc.py $O/tcc.log $O/tcc/*/*_counter_collection.csv | tee $O/tcc_summary.txt
python3 $R/tools/exp_r4_breakdown_pmc.py $O/wr.log $O/wr/*/*_counter_collection.csv | tee $O/wr_summary.txt
