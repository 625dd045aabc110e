#!/usr/bin/env python3
"""Repeat the headline workload many times and check every launch: the kernel's internal error
counter must stay 0 and the grid increment must be the same (to rounding) each time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch, bench, gridhip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, N, W, Q, S = bench.WORKLOADS["cfg3"]
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
gcf = bench.synth_kernels(W, Q, S, dev)
u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
ref = None
nbad = 0
for i in range(reps):
    G.zero_()
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    err = ctx.get_option("errors")
    s = G.sum().item(); a = G.abs().sum().item()
    if ref is None:
        ref = (s, a); Gref = G.clone()
    d = abs(s - ref[0]) / ref[1]
    dm = (G - Gref).abs()
    mx = dm.max().item()
    if mx > 1e-9:
        idx = dm.argmax().item(); nb = int((dm > 1e-9).sum().item())
        ys, xs = torch.nonzero(dm > 1e-9, as_tuple=True)
        print(f"  ** launch {i}: max|dG|={mx:.3e} at (y={idx // N}, x={idx % N}); {nb} cells differ; "
              f"y range {ys.min().item()}..{ys.max().item()}, x range {xs.min().item()}..{xs.max().item()}", flush=True)
    print(f"launch {i}: errors={err} checksum_dev={d:.2e} abs={a:.6e}", flush=True)
    if not (err == 0 and d < 1e-12):
        nbad += 1
        print(f"  ** MISMATCH at launch {i}: errors={err} dev={d:.3e}", flush=True)
print("stress ok" if nbad == 0 else f"stress FAILED: {nbad} bad launches")
sys.exit(1 if nbad else 0)
