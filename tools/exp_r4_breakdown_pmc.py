"""Attribute the k_primary dispatches of tools/exp_r4_breakdown.py under rocprofv3 --pmc to its variants (by order) and print
the counters' means per launch:  python3 tools/exp_r4_breakdown_pmc.py <run.log> <counter_collection.csv> [...]"""
import collections, csv, re, sys
log = sys.argv[1]
variants = [(m.group(1), m.group(2), int(m.group(3)), float(m.group(4)))
            for m in re.finditer(r"VARIANT (\S+) (\S+) launches=(\d+) geometry_us=([\d.]+)", open(log).read())]
for f in sys.argv[2:]:
    rows = collections.defaultdict(dict)                      # dispatch id -> counter -> value (k_primary only)
    for r in csv.DictReader(open(f)):
        if "k_primary" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    pos = 0
    for scene, name, n, us in variants:
        chunk = ids[pos:pos + n]; pos += n
        if not chunk:
            break
        keys = sorted(rows[chunk[0]])
        use = chunk[1:] if len(chunk) > 1 else chunk            # (the first launch of a variant also pays scene-side lazies)
        print(scene, name, f"{us:.1f}us", {k: round(sum(rows[i][k] for i in use) / len(use) / 1e6, 3) for k in keys}, "M per launch")
