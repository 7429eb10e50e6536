"""gpurun_out/r4prof (tools/profile_r4.sh) -> profiles/r04_*: the kernel-stats CSVs, the bench line taken under rocprof, ONE PMC summary
per workload (the dispatches of a pass are attributed to tools/run_configs_r4.py's workloads by order), and profiles/r04_claims.md:
every number DESIGN.md 7 quotes, with the file and row it comes from.  HBM bytes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE from passes of their own, in KiB; FETCH_SIZE doubled on gfx950 for wide coalesced streams -- both figures are kept,
K1's reads are byte gathers and the x2 rule is calibrated on 16-B-per-lane streams."""
import collections, csv, glob, json, os, re, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16, HBM_PEAK_GBS, N_SIMD, CLOCK_HZ, VALU_CYCLES
R = "gpurun_out/r4prof"
os.makedirs("profiles", exist_ok=True)
claims = []


def newest(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None


def stats(d, out):
    f = newest(f"{R}/{d}/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, out)
    return {r["Name"]: r for r in csv.DictReader(open(out))} if f else {}


def hbm(c):
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f, w = c["FETCH_SIZE"]["mean_per_launch"] * 1024, c["WRITE_SIZE"]["mean_per_launch"] * 1024
        return {"fetch_raw": f, "fetch_gfx950_x2": 2 * f, "write": w, "total_raw_fetch": f + w, "total_guide_rule": 2 * f + w}
    return None


def rows_of(d, match):
    """dispatch id -> {counter: value} of the kernels `match` accepts, in dispatch order"""
    f = newest(f"{R}/{d}/*/*_counter_collection.csv")
    rows = collections.defaultdict(dict)
    if f:
        for r in csv.DictReader(open(f)):
            if match(r["Kernel_Name"]):
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                rows[int(r["Dispatch_Id"])]["__name"] = r["Kernel_Name"]
    return [rows[k] for k in sorted(rows)]


def mean_counters(rs):
    agg = collections.defaultdict(list)
    for r in rs:
        for k, v in r.items():
            if k != "__name":
                agg[k].append(v)
    return {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in agg.items()}


# ---- the bench line ------------------------------------------------------------------------------------------------------
st_bench = stats("trace", "profiles/r04_bench_n1_kernel_stats.csv")
st_cfg = stats("cfg_trace", "profiles/r04_configs_kernel_stats.csv")
fpl = 128
if os.path.exists(f"{R}/bench_under_rocprof.json"):
    shutil.copy(f"{R}/bench_under_rocprof.json", "profiles/r04_bench_n1_under_rocprof.json")
    try:
        fpl = json.load(open(f"{R}/bench_under_rocprof.json"))["roofline"]["frames_per_launch"]
    except Exception:
        pass
k1, tg = {}, {}
for p in ("fetch", "write", "sq", "misc"):
    k1.update(mean_counters(rows_of(f"k1_{p}", lambda n: "k_primary<7" in n and ", 1, true" in n)))
    tg.update(mean_counters(rows_of(f"k1_{p}", lambda n: "k_tile_tags<true>" in n)))
hb = hbm(k1)
if hb and hbm(tg):                       # one k_tile_tags launch runs ahead of every K1 launch: its traffic belongs to the step
    for k, v in hbm(tg).items():
        hb[k] += v
json.dump({"kernel": "k_primary<7 (DF, hand-written look-up loop), false, 1 (primary only), true (slot table)> + k_tile_tags<true> ahead of it",
           "frames_per_launch": fpl, "csrc_sha16": csrc_sha16(), "counters": k1, "counters_k_tile_tags": tg, "hbm_bytes_per_launch": hb},
          open("profiles/r04_k_primary_pmc.json", "w"), indent=1)

# ---- one summary per workload ----------------------------------------------------------------------------------------------
def variants(log):
    return [(m.group(1), int(m.group(2)), int(m.group(3))) for m in re.finditer(r"VARIANT (\S+) geometry=(\d+) denoise_passes=(\d+)", open(log).read())] if os.path.exists(log) else []


per = collections.defaultdict(lambda: {"geometry": {}, "denoise": [{}, {}]})
for p in ("fetch", "write", "sq", "misc"):
    vs = variants(f"{R}/cfg_{p}.log")
    geo = rows_of(f"cfg_{p}", lambda n: "k_primary<" in n)
    den = rows_of(f"cfg_{p}", lambda n: "k_denoise" in n)
    gi = di = 0
    for tag, n, passes in vs:
        g = geo[gi:gi + n]; gi += n
        per[tag]["geometry"].update(mean_counters(g[1:] if len(g) > 1 else g))        # (a workload's first launch also pays its lazies)
        per[tag]["kernel"] = g[0]["__name"] if g else None
        d = den[di:di + n * passes]; di += n * passes
        for q in range(passes):
            dq = d[q::passes]
            per[tag]["denoise"][q].update(mean_counters(dq[1:] if len(dq) > 1 else dq))
            per[tag].setdefault("denoise_kernels", [None, None])[q] = dq[0]["__name"] if dq else None
desc = {"config3": "BASELINE configs[2]: treehouse 256^3, 1920x1080, primary + shadow ray, one frame per launch",
        "defaults": "the reference's defaults: treehouse 256^3, 1920x1080, AO 4 x 64, shadow ray, <= 5 bounces, one frame per launch",
        "mandelbulb": "BASELINE configs[3]: Mandelbulb 512^3, 3840x2160, 2 bounces, AO 4, shadow ray, one frame per launch",
        "brick": "BASELINE configs[4]: 2048^3 brick scene, 3840x2160, max_steps 6144, 4 bounces, AO 4, one frame per launch"}
for tag, d in per.items():
    if tag in desc:
        json.dump({"workload": desc[tag], "kernel": d.get("kernel"), "csrc_sha16": csrc_sha16(), "counters": d["geometry"], "hbm_bytes_per_launch": hbm(d["geometry"])},
                  open(f"profiles/r04_{tag}_pmc.json", "w"), indent=1)
    names = {"config3": ("k_denoise_pass0", "k_denoise"), "config3_fast": (None, "k_denoise_fast"), "config3_literal": ("k_denoise_pass0_literal", "k_denoise_literal")}
    for q, nm in enumerate(names.get(tag, (None, None))):
        if nm and d["denoise"][q]:
            json.dump({"workload": f"{tag}: denoiser pass {q} at 1920x1080", "kernel": (d.get("denoise_kernels") or [None, None])[q], "csrc_sha16": csrc_sha16(),
                       "counters": d["denoise"][q], "hbm_bytes_per_launch": hbm(d["denoise"][q])}, open(f"profiles/r04_{nm}_pmc.json", "w"), indent=1)

# ---- the claims table ---------------------------------------------------------------------------------------------------------
def us(row):
    return float(row["AverageNs"]) / 1e3


lines = ["# Round-4 numbers and where each comes from (generated by tools/profile_r4_collect.py; csrc digest " + csrc_sha16() + ")", "",
         "| claim | value | file | row / field |", "|---|---|---|---|"]
for name, row in st_bench.items():
    if "k_primary<7" in name and ", 1, true" in name or "k_tile_tags<true>" in name:
        lines.append(f"| bench line, kernel time per {fpl}-frame launch | {us(row):.1f} us avg over {row['Calls']} launches (min {float(row['MinNs']) / 1e3:.1f}) | profiles/r04_bench_n1_kernel_stats.csv | `{name[:70]}` |")
try:
    bj = json.load(open("profiles/r04_bench_n1_under_rocprof.json"))
    rf = bj["roofline"]
    for k in ("kernel_ms", "frac", "frac_requested", "hbm_frac", "frac_reference_steps", "requested_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic"):
        lines.append(f"| bench line under rocprofv3: roofline.{k} | {rf.get(k)} | profiles/r04_bench_n1_under_rocprof.json | roofline.{k} |")
    lines.append(f"| bench line under rocprofv3: value | {bj['value']} Mrays/s | profiles/r04_bench_n1_under_rocprof.json | value |")
except Exception as e:
    lines.append(f"| bench line under rocprofv3 | missing: {e} | | |")
if hb:
    v = k1.get("SQ_INSTS_VALU", {}).get("mean_per_launch")
    lines.append(f"| K1 + tags HBM bytes per launch (guide rule / raw fetch) | {hb['total_guide_rule'] / 1e9:.3f} / {hb['total_raw_fetch'] / 1e9:.3f} GB | profiles/r04_k_primary_pmc.json | hbm_bytes_per_launch |")
    if v:
        lines.append(f"| K1 vector instructions per launch / per frame; issue time at 4 cycles | {v / 1e6:.1f} M / {v / fpl / 1e6:.2f} M; {v * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e3:.3f} ms | profiles/r04_k_primary_pmc.json | counters.SQ_INSTS_VALU |")
plain = {m.group(1): (float(m.group(3)), float(m.group(4)) if int(m.group(2)) > 0 else -1.0)
         for m in re.finditer(r"VARIANT (\S+) geometry=\d+ denoise_passes=(\d+) geometry_us=([\d.]+) denoise_us=([-\d.]+)", open(f"{R}/cfg_plain.log").read())} if os.path.exists(f"{R}/cfg_plain.log") else {}
for tag in ("config3", "defaults", "mandelbulb", "brick"):
    d = per.get(tag)
    if not d:
        continue
    h, c = hbm(d["geometry"]), d["geometry"]
    t = plain.get(tag, (None, None))[0]
    lines.append(f"| {tag}: geometry, one frame (library events, no profiler) | {t} us | gpurun_out/r4prof/cfg_plain.log (copied: profiles/r04_configs_plain.log) | VARIANT {tag} |")
    if h:
        lines.append(f"| {tag}: HBM bytes per frame (guide rule / raw fetch / written) | {h['total_guide_rule'] / 1e6:.1f} / {h['total_raw_fetch'] / 1e6:.1f} / {h['write'] / 1e6:.1f} MB | profiles/r04_{tag}_pmc.json | hbm_bytes_per_launch |")
    if "SQ_INSTS_VALU" in c:
        v = c["SQ_INSTS_VALU"]["mean_per_launch"]
        lines.append(f"| {tag}: vector instructions per frame; issue time | {v / 1e6:.1f} M; {v * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e6:.1f} us | profiles/r04_{tag}_pmc.json | counters.SQ_INSTS_VALU |")
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        lines.append(f"| {tag}: SQ_WAIT_ANY / SQ_WAVE_CYCLES; L2 hit | {c['SQ_WAIT_ANY']['mean_per_launch'] / c['SQ_WAVE_CYCLES']['mean_per_launch']:.3f}; "
                     f"{(c['TCC_HIT_sum']['mean_per_launch'] / (c['TCC_HIT_sum']['mean_per_launch'] + c['TCC_MISS_sum']['mean_per_launch'])) if 'TCC_HIT_sum' in c else float('nan'):.3f} | profiles/r04_{tag}_pmc.json | counters |")
for name, row in st_cfg.items():
    if "k_denoise" in name or "k_primary" in name:
        lines.append(f"| configurations under rocprofv3 --kernel-trace: `{name[:60]}` | {us(row):.1f} us avg, {float(row['MinNs']) / 1e3:.1f} min, {row['Calls']} calls | profiles/r04_configs_kernel_stats.csv | same |")
K3_BYTES = {"k_denoise": 28, "k_denoise_fast": 28, "k_denoise_literal": 28, "k_denoise_pass0": 8, "k_denoise_pass0_literal": 8}     # bytes per pixel and pass
for nm, bpp in K3_BYTES.items():
    f = f"profiles/r04_{nm}_pmc.json"
    if not os.path.exists(f):
        continue
    j = json.load(open(f))
    if j.get("csrc_sha16") != csrc_sha16():
        continue
    h, c = j.get("hbm_bytes_per_launch"), j["counters"]
    alg = 1920 * 1080 * bpp
    kn = (j.get("kernel") or "")[:48]
    if h:
        lines.append(f"| K3 `{kn}`: HBM bytes per pass (guide rule / raw fetch / written) against {alg / 1e6:.2f} MB by the accounting | {h['total_guide_rule'] / 1e6:.1f} / {h['total_raw_fetch'] / 1e6:.1f} / {h['write'] / 1e6:.1f} MB = "
                     f"{h['total_guide_rule'] / alg:.2f}x / {h['total_raw_fetch'] / alg:.2f}x | {f} | hbm_bytes_per_launch |")
    if "SQ_INSTS_VALU" in c:
        v = c["SQ_INSTS_VALU"]["mean_per_launch"]
        lines.append(f"| K3 `{kn}`: vector instructions per pass; issue time | {v / 1e6:.2f} M; {v * VALU_CYCLES / N_SIMD / CLOCK_HZ * 1e6:.1f} us | {f} | counters.SQ_INSTS_VALU |")
for name, row in st_cfg.items():
    if "k_denoise_pair<true, 3>" in name:
        lines.append(f"| K3 weighted pass at 1080p, exact output: accounting bytes / kernel time / HBM peak | {1920 * 1080 * 28 / 1e6:.2f} MB / {us(row):.1f} us = {1920 * 1080 * 28 / (us(row) * 1e-6) / 8e12:.3f} of 8 TB/s | profiles/r04_configs_kernel_stats.csv | `{name[:50]}` |")
for tag, (g, dn) in plain.items():
    if dn > 0:
        lines.append(f"| {tag}: two denoiser passes, library events, no profiler | {dn} us | profiles/r04_configs_plain.log | VARIANT {tag} denoise_us |")
if os.path.exists(f"{R}/cfg_plain.log"):
    shutil.copy(f"{R}/cfg_plain.log", "profiles/r04_configs_plain.log")
open("profiles/r04_claims.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
# DESIGN.md 7 carries the same table between two markers: this script is the only writer of the numbers there
if os.path.exists("DESIGN.md"):
    d = open("DESIGN.md").read()
    b, e = "<!-- r04_claims:begin -->", "<!-- r04_claims:end -->"
    if b in d and e in d:
        d = d[:d.index(b) + len(b)] + "\n" + "\n".join(lines[2:]) + "\n" + d[d.index(e):]
        open("DESIGN.md", "w").write(d)
