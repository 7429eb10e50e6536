#!/bin/bash
# SQ counters of K1 (tools/exp_k1.py) with the context options given as arguments, e.g. tools/pmc_k1.sh tag sky_fast=0
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/pmc_k1_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_SMEM \
    -d $O/sq --output-format csv -- python3 $R/tools/exp_k1.py "$@" --reps 6 > $O/run.log 2> $O/err.log
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
for f in glob.glob(o + "/sq/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_primary" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: round(sum(v) / len(v) / 64) for k, v in agg.items()}, "per frame,", len(next(iter(agg.values()))), "launches")
PY
