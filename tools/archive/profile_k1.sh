#!/bin/bash
# Profile bench.py's K1 under rocprofv3 on the GPU box: one --kernel-trace --stats run and four --pmc passes
# (FETCH_SIZE and WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md prescribes; never combined with a trace).
# Usage (through gpurun):  bash tools/profile_k1.sh
# Afterwards, here:        python tools/pmc_summary.py <tag> "k_primary<4" 32 (frames per launch); copy the stats csv into profiles/.  Remove gpurun_out/pmc_* and gpurun_out/prof here first.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_misc
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof --output-format csv -- $B --steps 200 --warmup 20 \
    > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d $R/gpurun_out/pmc_fetch --output-format csv -- $B --steps 20 --warmup 5 \
    > /dev/null 2> $R/gpurun_out/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum -d $R/gpurun_out/pmc_write --output-format csv -- $B --steps 20 --warmup 5 \
    > /dev/null 2> $R/gpurun_out/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    -d $R/gpurun_out/pmc_sq --output-format csv -- $B --steps 20 --warmup 5 > /dev/null 2> $R/gpurun_out/pmc_sq.err
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAVES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum \
    -d $R/gpurun_out/pmc_misc --output-format csv -- $B --steps 20 --warmup 5 > /dev/null 2> $R/gpurun_out/pmc_misc.err
tail -c 400 $R/gpurun_out/prof_bench.json
