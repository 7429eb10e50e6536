#!/bin/bash
# GRBM_GUI_ACTIVE (GPU clock cycles while a kernel runs) of one kernel of a tool script: tools/pmc_clk.sh KERNEL_SUBSTRING script.py
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; S=$2
O=$R/gpurun_out/pmc_clk
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES -d $O/c --output-format csv -- python3 $R/$S > $O/run.log 2> $O/err.log
python3 - "$O" "$K" <<'PY'
import csv, glob, sys, collections
o, k = sys.argv[1], sys.argv[2]
for f in glob.glob(o + "/c/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({n: round(sum(v) / len(v)) for n, v in agg.items()}, len(next(iter(agg.values()), [])), "launches")
PY
tail -3 $O/err.log
