#!/usr/bin/env python3
"""Per-phase cycle breakdown of the sorted tile kernel (option dbg=16: thread 0 of every work-group stamps
clock64 at each barrier).  usage: python tools/phase_profile.py [--workload=cfg3] [opt=val,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the "dbg" option and the stamping instantiation exist in the tuning build only (make -C csrc tuning)
os.environ.setdefault("GRIDHIP_LIB", os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "lib", "libgridhip_tuning.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch
import bench
import gridhip

sets = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")] or [""]
wl = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--workload=")]
n, N, W, Q, S = bench.WORKLOADS[wl[0] if wl else "cfg3"]
for a_ in sys.argv[1:]:  # --n= / --grid= / --planes= override the workload's shape (the stamps exist for 15 x 15 only)
    if a_.startswith("--n="):
        n = int(float(a_[4:]))
    if a_.startswith("--grid="):
        N = int(a_[7:])
    if a_.startswith("--planes="):
        W = int(a_[9:])
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
gcf = bench.synth_kernels(W, Q, S, dev)
u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
ctx.enable_timing(True)
names = ["work fetch + waiting at barriers", "histogram", "scan", "scatter + value gather", "accumulate walk", "waiting for the slowest wave", "flush + clear"]
for s in sets:
    for kv in filter(None, s.split(",")):
        k, val = kv.split("=")
        ctx.set_option(k, int(val))
    if "dbg=" not in s:
        ctx.set_option("dbg", 16)
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    t = ctx.last_timing()
    cyc = [ctx.get_option(f"prof{i}") for i in range(7)]
    # phases 0..3 are stamped by the sorter wave, 4..6 by walker wave 0: each role's stamps add up to the
    # work-group's whole time
    for role, idx in (("sorter wave", (0, 1, 2, 3)), ("walker wave 0", (4, 5, 6))):
        tot = sum(cyc[i] for i in idx) or 1
        print(f"[{s or 'default'}] kernel {t[2]:.2f} ms; {role}: share of its time per phase")
        for i in idx:
            print(f"  {names[i]:28s} {100.0 * cyc[i] / tot:6.2f} %   ({t[2] * cyc[i] / tot:6.2f} ms)")
    wv = [ctx.get_option(f"prof{8 + i}") for i in range(16)]
    print("  walk time per wave (relative to the slowest): " + " ".join(f"{x / max(wv):.2f}" for x in wv))
