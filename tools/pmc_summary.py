"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_*/) for the K1 kernel into profiles/<tag>_k_primary_pmc.json.

HBM bytes per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE come from separate passes
(TCC slots), are in KiB, and FETCH_SIZE is doubled on gfx950 for wide coalesced streams.  K1's reads are byte
gathers, not wide streams, so both the raw and the doubled figure are recorded and the raw one is cross-checked
against TCC_EA0_RDREQ * 64 B."""
import csv, glob, json, os, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16          # digest of the kernel sources: bench.py refuses a summary collected on other sources

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_primary<4"
frames_per_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
out = {}
for d in sorted(glob.glob("gpurun_out/pmc_*/")):
    files = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:               # gpurun merges every run into the same directory: the newest file is this run's
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
res = {"kernel": kernel, "frames_per_launch": frames_per_launch, "csrc_sha16": csrc_sha16(), "counters": out}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    f, w = out["FETCH_SIZE"]["mean_per_launch"] * 1024, out["WRITE_SIZE"]["mean_per_launch"] * 1024
    res["hbm_bytes_per_launch"] = {"fetch_raw": f, "fetch_gfx950_x2": 2 * f, "write": w,
                                   "total_raw_fetch": f + w, "total_guide_rule": 2 * f + w}
    if "TCC_EA0_RDREQ_sum" in out:
        res["hbm_bytes_per_launch"]["fetch_from_rdreq_x64B"] = out["TCC_EA0_RDREQ_sum"]["mean_per_launch"] * 64
    if "TCC_EA0_WRREQ_sum" in out:
        res["hbm_bytes_per_launch"]["write_from_wrreq_x64B"] = out["TCC_EA0_WRREQ_sum"]["mean_per_launch"] * 64
os.makedirs("profiles", exist_ok=True)
path = f"profiles/{tag}_{sys.argv[4] if len(sys.argv) > 4 else 'k_primary'}_pmc.json"
json.dump(res, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
