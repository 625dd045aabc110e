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
for i in range(reps):
    G.zero_()
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    err = ctx.get_option("errors")
    s = G.sum().item(); a = G.abs().sum().item()
    if ref is None: ref = (s, a)
    d = abs(s - ref[0]) / ref[1]
    print(f"launch {i}: errors={err} checksum_dev={d:.2e} abs={a:.6e}", flush=True)
    assert err == 0 and d < 1e-12
print("stress ok")
