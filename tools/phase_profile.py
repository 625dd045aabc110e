#!/usr/bin/env python3
"""Per-phase cycle breakdown of the sorted tile kernel (option dbg=16: thread 0 of every work-group stamps
clock64 at each barrier).  usage: python tools/phase_profile.py [--workload cfg3] [opt=val,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch
import bench
import gridhip

sets = [a for a in sys.argv[1:] if "=" in a] or [""]
n, N, W, Q, S = bench.WORKLOADS["cfg3"]
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
gcf = bench.synth_kernels(W, Q, S, dev)
u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
ctx.enable_timing(True)
names = ["fetch+init/wait", "histogram", "scan", "scatter+gather", "accumulate(wave0)", "wait slowest", "flush"]
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
    tot = sum(cyc)
    print(f"[{s or 'default'}] kernel {t[2]:.2f} ms; share of work-group time per phase:")
    for nm, c in zip(names, cyc):
        print(f"  {nm:20s} {100.0 * c / tot:6.2f} %   ({t[2] * c / tot:6.2f} ms)")
    wv = [ctx.get_option(f"prof{8 + i}") for i in range(16)]
    print("  walk time per wave (relative to the slowest): " + " ".join(f"{x / max(wv):.2f}" for x in wv))
