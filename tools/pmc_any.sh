#!/bin/bash
# SQ / cache counters of one kernel of any command: tools/pmc_any.sh TAG KERNEL_SUBSTRING python3 script.py args...
# (two --pmc passes, never together with a trace; prints the means per launch)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; K=$2; shift; shift
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_SMEM \
    -d $O/sq --output-format csv -- "$@" > $O/run1.log 2> $O/err1.log
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_LDS SQ_ACTIVE_INST_VALU \
    -d $O/mem --output-format csv -- "$@" > $O/run2.log 2> $O/err2.log
python3 - "$O" "$K" <<'PY'
import csv, glob, sys, collections
o, k = sys.argv[1], sys.argv[2]
for d in ("sq", "mem"):
    for f in glob.glob(o + "/" + d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if k in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print({n: round(sum(v) / len(v)) for n, v in agg.items()}, "per launch,", len(next(iter(agg.values()), [])), "launches")
PY
