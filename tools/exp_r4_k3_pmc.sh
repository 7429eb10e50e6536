#!/bin/bash
# K3 counters: the two denoiser passes of config 3 (tools/archive/exp_k3_pmc.py: 6 frames exact + 6 VRT_DENOISE_FAST) under two PMC
# passes, with round 4's kernels (k_denoise_p0 + k_denoise_pair, the default) and with VRT_DENOISE_PAIR=0 VRT_DENOISE_P0=0 (k_denoise_ver for
# both passes); per-kernel means printed by the python below.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4k3pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pair in 1 0; do
  export VRT_DENOISE_PAIR=$pair VRT_DENOISE_P0=$pair      # (pair=0: round 3's kernels for both passes, k_denoise_ver)
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq$pair --output-format csv -- python3 $R/tools/archive/exp_k3_pmc.py > $O/sq$pair.log 2> $O/sq$pair.err
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU -d $O/lds$pair --output-format csv -- python3 $R/tools/archive/exp_k3_pmc.py > $O/lds$pair.log 2> $O/lds$pair.err
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/tr$pair --output-format csv -- python3 $R/tools/archive/exp_k3_pmc.py > $O/tr$pair.log 2> $O/tr$pair.err
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4k3pmc"
for pair in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("sq", "lds"):
        for f in glob.glob(f"{O}/{d}{pair}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "k_denoise" in r["Kernel_Name"]:
                    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(f"pair={pair} {k}: " + ", ".join(f"{c}={sum(v) / len(v) / 1e6:.3f}M" for c, v in sorted(cs.items())))
    for f in glob.glob(f"{O}/tr{pair}/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "k_denoise" in r["Name"]:
                print(f"pair={pair} trace {r['Name'][:60]}: calls={r['Calls']} avg_ns={r['AverageNs']} min_ns={r['MinNs']}")
PY
