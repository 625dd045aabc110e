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
