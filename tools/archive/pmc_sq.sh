#!/bin/bash
# SQ counters of bench.py's K1 (one --pmc pass, no trace): bash tools/pmc_sq.sh <outdir-under-gpurun_out> [env assignments...]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/$1; shift
rm -rf $D
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    -d $D --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > /dev/null 2> $D.err
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$D/*/*_counter_collection.csv"))[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_primary" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:24s} {sum(v)/len(v)/64/1e6:10.3f} M per frame  ({len(v)} launches)")
PY
