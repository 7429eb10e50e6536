#!/bin/bash
# kernel-trace statistics of any tool script: tools/trace_any.sh TAG script.py [args...]   (through gpurun)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
S=$1; shift
O=$R/gpurun_out/trace_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/t --output-format csv -- python3 $R/$S "$@" > $O/run.log 2> $O/err.log
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
for f in glob.glob(o + "/t/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.2f} us  min {float(r["MinNs"])/1e3:9.2f}  max {float(r["MaxNs"])/1e3:9.2f}')
PY
tail -2 $O/run.log
