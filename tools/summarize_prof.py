#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into a small markdown summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
short = lambda n: n.split("(")[0].replace("void ", "").replace("gridhip::", "")[:48]

print(f"# rocprofv3 summary: {out}\n")
for f in glob.glob(os.path.join(out, "*trace.log")):
    for line in open(f):
        if line.startswith("{"):
            print("bench line under the profiler (kernel-trace pass):\n```\n" + line.strip() + "\n```\n")

stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    print("## kernel-trace --stats (gridhip kernels only)\n")
    print("| kernel | calls | avg ms | min ms | max ms | % of GPU time |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(stats[0])):
        if "gridhip" in r["Name"]:
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['MinNs'])/1e6:.3f} | "
                  f"{float(r['MaxNs'])/1e6:.3f} | {r['Percentage']} |")
    print()

print("## PMC passes (per dispatch, averaged over the dispatches of each gridhip kernel)\n")
print("| kernel | counter | dispatches | mean | note |")
print("|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "gridhip" not in name:
            continue
        acc[(short(name), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        mean = sum(v) / len(v)
        note = ""
        if c == "FETCH_SIZE":
            note = f"KB; x1024 = {mean*1024/1e9:.3f} GB; x2 (gfx950 wide-read correction) = {mean*2048/1e9:.3f} GB"
        if c == "WRITE_SIZE":
            note = f"KB; x1024 = {mean*1024/1e9:.3f} GB"
        print(f"| {k} | {c} | {len(v)} | {mean:.6g} | {note} |")


# ---- traffic.json: HBM bytes per launch of the tile kernel, stamped with the kernel sources it was measured on
import json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

per = {}
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name, c = r.get("Kernel_Name", ""), r["Counter_Name"]
        if "gridhip" in name and c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"):
            per.setdefault((short(name), c), []).append(float(r["Counter_Value"]))
mean = lambda k, c: (sum(per[(k, c)]) / len(per[(k, c)])) if (k, c) in per else None
tile = [k for (k, c) in per if k.startswith("tile_grid")]
if tile:
    k = sorted(set(tile))[0]
    fetch, write = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    rec = {"kernel": k, "csrc_sha16": bench.csrc_fingerprint(),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x 2 (gfx950 wide-read "
                     "correction, MI355X_MICROARCH.md) + WRITE_SIZE, KB -> bytes"}
    if fetch is not None and write is not None:
        rec["fetch_bytes"] = fetch * 2048.0
        rec["write_bytes"] = write * 1024.0
        rec["hbm_bytes_per_launch"] = fetch * 2048.0 + write * 1024.0
    cal = [kk for kk in set(x for (x, _) in per) if kk.startswith("bin_count")]
    if cal and mean(cal[0], "FETCH_SIZE") is not None:
        rec["calibration_bin_count_fetch_bytes_x2"] = mean(cal[0], "FETCH_SIZE") * 2048.0
        rec["calibration_note"] = "bin_count_kernel reads u, v, wbin once: 24 B per visibility"
    h, m = mean(k, "TCC_HIT_sum"), mean(k, "TCC_MISS_sum")
    if h is not None and m is not None:
        rec["tcc_hit_rate"] = h / (h + m)
    g = mean(k, "GRBM_GUI_ACTIVE")
    if g is not None:
        rec["grbm_gui_active_sum_over_xcds"] = g
    json.dump(rec, open(os.path.join(out, "traffic_fragment.json"), "w"), indent=1)
    print("\n## traffic fragment (merge into profiles/traffic.json under the workload's key)\n```\n" + json.dumps(rec, indent=1) + "\n```")
