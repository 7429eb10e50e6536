"""gpurun_out/r3prof (tools/profile_r3.sh) -> profiles/r03_*: kernel-stats CSVs, the bench line taken under rocprof, and PMC
summaries per kernel.  HBM bytes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE from passes of their own, in KiB;
FETCH_SIZE doubled on gfx950 for wide coalesced streams (both figures are kept: K1's reads are byte gathers)."""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import csrc_sha16
R = "gpurun_out/r3prof"
os.makedirs("profiles", exist_ok=True)

def newest(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None

def stats(d, out):
    f = newest(f"{R}/{d}/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, out)
    return f

def counters(dirs, match, per=1.0):
    res = {}
    for d in dirs:
        f = newest(f"{R}/{d}/*/*_counter_collection.csv")
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match(r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            res[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    return res

def hbm(c):
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f, w = c["FETCH_SIZE"]["mean_per_launch"] * 1024, c["WRITE_SIZE"]["mean_per_launch"] * 1024
        return {"fetch_raw": f, "fetch_gfx950_x2": 2 * f, "write": w, "total_raw_fetch": f + w, "total_guide_rule": 2 * f + w}
    return None

stats("trace", "profiles/r03_bench_n1_kernel_stats.csv")
stats("cfg_trace", "profiles/r03_configs_kernel_stats.csv")
if os.path.exists(f"{R}/bench_under_rocprof.json"):
    shutil.copy(f"{R}/bench_under_rocprof.json", "profiles/r03_bench_n1_under_rocprof.json")
fpl = 64
try:
    fpl = json.load(open(f"{R}/bench_under_rocprof.json"))["roofline"]["frames_per_launch"]
except Exception:
    pass
# K1 of the headline: the DF kernel through the hand-written loop (template argument 7), launches of the whole batch only
k1 = counters(["k1_fetch", "k1_write", "k1_sq", "k1_misc"], lambda n: "k_primary<7" in n and ", 1, true" in n)
tg = counters(["k1_fetch", "k1_write", "k1_sq", "k1_misc"], lambda n: "k_tile_tags<true>" in n)
hb = hbm(k1)
if hb and hbm(tg):                       # one k_tile_tags launch runs ahead of every K1 launch: its traffic belongs to the step
    for k, v in hbm(tg).items():
        hb[k] += v
out = {"kernel": "k_primary<7 (DF, hand-written look-up loop), false, 1 (primary only), true (slot table)> + k_tile_tags<true> ahead of it",
       "frames_per_launch": fpl, "csrc_sha16": csrc_sha16(), "counters": k1, "counters_k_tile_tags": tg, "hbm_bytes_per_launch": hb}
json.dump(out, open("profiles/r03_k_primary_pmc.json", "w"), indent=1)
print("K1:", {k: round(v["mean_per_launch"] / fpl / 1e6, 3) for k, v in k1.items()}, "M per frame;", out["hbm_bytes_per_launch"])
sets = {
    "r03_megakernel_pmc.json": ("k_primary<7, false, 4 (megakernel; the scene has no metallic voxel, so the form without the bounce loop)>: config 3 (shadow ray; launches 1-20) and the reference defaults (AO 4, shadow, <= 5 bounces; launches 21-30), 1080p, one frame per launch",
                                lambda n: "k_primary<7" in n and (", 2, false" in n or ", 4, false" in n)),
    "r03_k_denoise_pmc.json": ("k_denoise_ver<false, true, 3, false> (the verified weighted pass: exact output, the default), 1080p", lambda n: "k_denoise_ver<false, true, 3, false>" in n),
    "r03_k_denoise_fast_pmc.json": ("k_denoise_ver<false, false, 3, false> (VRT_DENOISE_FAST weighted pass: the verified pass' cheap half alone), 1080p", lambda n: "k_denoise_ver<false, false, 3, false>" in n),
    "r03_k_denoise_pass0_pmc.json": ("k_denoise_ver<false, true, 1, true> (pass 0: plain blur, verified form), 1080p", lambda n: "k_denoise_ver<false, true, 1, true>" in n),
    "r03_k_denoise_exact_pmc.json": ("k_denoise_lds<false, false, false, true> (the exact weighted pass computing every pixel, context option denoise_verified = 0: what the verified pass replaced), 1080p", lambda n: "k_denoise_lds<false, false, false, true>" in n),
    "r03_mandelbulb_pmc.json": ("k_primary<7, false, 2 (megakernel)>: BASELINE configs[3], Mandelbulb 512^3, 3840x2160, 2 bounces, AO 4, shadow ray, one frame per launch (the last five launches of the run; the counters below average ALL launches of the megakernel with the bounce loop, of which the 1080p reference defaults are the first ten)",
                                lambda n: "k_primary<7" in n and ", 2, false" in n),
    "r03_brick_pmc.json": ("k_primary<6 (BRICK), false, 2 (megakernel)>: BASELINE configs[4], 2048^3 brick scene, 3840x2160, max_steps 6144, 4 bounces, AO 4, one frame per launch",
                           lambda n: "k_primary<6" in n),
}
for name, (desc, m) in sets.items():
    c = counters(["cfg_fetch", "cfg_write", "cfg_sq", "cfg_misc"], m)
    json.dump({"kernel": desc, "csrc_sha16": csrc_sha16(), "counters": c, "hbm_bytes_per_launch": hbm(c)}, open("profiles/" + name, "w"), indent=1)
    print(name, {k: round(v["mean_per_launch"] / 1e6, 3) for k, v in c.items() if k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "FETCH_SIZE", "WRITE_SIZE")}, hbm(c) and round(hbm(c)["total_guide_rule"] / 1e6, 1), "MB")
