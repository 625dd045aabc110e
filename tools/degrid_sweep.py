#!/usr/bin/env python3
"""degrid2 at cfg3's shape for several supports (10^8 visibilities unless --nvis): whole call, pre-pass, kernel.
usage: python tools/degrid_sweep.py [--nvis N] S [S ...]   (option subfoot=1 via --subfoot)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import numpy as np
import torch
import bench
import gridhip

args = sys.argv[1:]
n, N, W, Q, _ = bench.WORKLOADS["cfg3"]
subfoot = 0
if "--subfoot" in args:
    args.remove("--subfoot")
    subfoot = 1
if "--nvis" in args:
    i = args.index("--nvis")
    n = int(float(args[i + 1]))
    del args[i:i + 2]
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
ctx.enable_timing(True)
ctx.set_option("subfoot", subfoot)
for S in [int(a) for a in args] or [15, 17, 21, 25, 31]:
    gcf = bench.synth_kernels(W, Q, S, dev)
    u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
    G = torch.randn((N, N), dtype=torch.float64, device=dev).to(torch.complex128)
    out = torch.empty(n, dtype=torch.complex128, device=dev)
    ctx.degrid2(gcf, G, (u, v, None), wb, out=out)
    ts = []
    for _ in range(3):
        ctx.degrid2(gcf, G, (u, v, None), wb, out=out)
        ts.append(ctx.last_timing())
    t = np.array(ts).min(axis=0)
    print(f"degrid2 {S}x{S} subfoot={subfoot}: total {t[0]:8.2f} ms  prepass {t[1]:6.2f}  kernel {t[2]:8.2f}  -> {n / t[0] / 1e3:8.1f} Mvis/s  "
          f"{S * S * n / t[2] / 1e6:7.0f} Gtaps/s (kernel)", flush=True)
    del gcf, u, v, wb, vis, G, out
