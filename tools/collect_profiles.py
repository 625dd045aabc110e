#!/usr/bin/env python3
"""Copy what tools/profile.sh and tools/phase_profile.py left under gpurun_out/ into profiles/ (tracked): the rocprofv3
summaries, the gridhip rows of the kernel statistics, the per-phase profiles, and profiles/traffic.json (the PMC traffic
bench.py quotes when its csrc_sha16 matches the kernel sources in the tree).
usage: python tools/collect_profiles.py <tag> <dir with phase_cfg3.log / phase_cfg5.log>     e.g.  r02 gpurun_out/r2x"""
import csv
import glob
import json
import os
import shutil
import sys

tag, phase_dir = sys.argv[1], sys.argv[2]
out = {}
for wl in ("cfg3", "cfg5"):
    src = f"gpurun_out/prof_{tag}_{wl}"
    out[wl] = json.load(open(f"{src}/traffic_fragment.json"))
    shutil.copy(f"{src}/summary.md", f"profiles/{tag}_{wl}_rocprofv3_summary.md")
    ks = max(glob.glob(f"{src}/trace/*/*kernel_stats.csv"), key=os.path.getmtime)
    rows = list(csv.reader(open(ks)))
    with open(f"profiles/{tag}_{wl}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "gridhip" in r[0]:
                w.writerow(r)
    t = open(f"{phase_dir}/phase_{wl}.log").read().split("\n", 1)[1]
    open(f"profiles/{tag}_phase_profile_{wl}.txt", "w").write(
        f"# tools/phase_profile.py --workload={wl} (tuning build), one MI355X\n" + t)
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
sys.path.insert(0, ".")
import bench
print("traffic.json csrc_sha16", out["cfg3"]["csrc_sha16"], "tree", bench.csrc_fingerprint())
for wl, o in out.items():
    print(wl, "fetch %.2f GB  write %.2f GB  total %.2f GB  TCC hit rate %.3f" % (
        o["fetch_bytes"] / 1e9, o["write_bytes"] / 1e9, o["hbm_bytes_per_launch"] / 1e9, o["tcc_hit_rate"]))
