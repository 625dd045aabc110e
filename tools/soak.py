#!/usr/bin/env python3
"""Soak run on one GPU: grid the bench workload many times and check every result by its checksum
(sum(G) == sum_k vis_k * sum_ij K[slice_k], every tap in range) - a race between the sorter and the walkers, a lost
record or a stale table would show as a step whose checksum is off.  Also alternates grid / degrid and option sets.
usage: python tools/soak.py [--workload cfg3] [--steps 300]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch
import bench
import gridhip

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg3")
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--support", type=int, default=0, help="another square support on the workload's shape (17 .. 32: parts of the tap list)")
ap.add_argument("--nvis", type=int, default=0)
a = ap.parse_args()
n, N, W, Q, S = bench.WORKLOADS[a.workload]
S = a.support or S
n = a.nvis or n
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
gcf = bench.synth_kernels(W, Q, S, dev)
u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)


def fc(p):
    x = N // 2 + p * N
    fl = torch.floor(x + 0.5 / Q)
    return torch.round((x - fl) * Q).clamp(0, Q - 1).long()


xf, yf = fc(u), fc(v)
ks = gcf.sum(dim=(3, 4))[wb, yf, xf]
expect = (vis * ks).sum()
scale = (vis.abs() * gcf.abs().sum(dim=(3, 4))[wb, yf, xf]).sum().item()
del ks
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
out = torch.empty(n, dtype=torch.complex128, device=dev)
sets = [{}, {"wgroups": 4}, {"chunk": 4096}, {"prepass": 6}, {"tile": 64}, {"wtable": 1}, {"bigtile": 1}, {"reserve_cus": 32}, {"yield_cus": 64},
        {"bigtile": 1, "wgroups": 4}]
worst, t0 = 0.0, time.time()
dref = None
for step in range(a.steps):
    opts = sets[step % len(sets)] if step % 5 == 4 else {}
    for k, val in opts.items():
        ctx.set_option(k, val)
    G.zero_()
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    err = abs((G.sum() - expect).item()) / scale
    worst = max(worst, err)
    bad = ctx.get_option("errors")
    if step % 10 == 0:  # degrid of the fresh grid: the same sample of predictions every time, up to the grid's rounding
        ctx.degrid2(gcf, G, (u, v, None), wb, out)
        s = out[:: max(1, n // 4096)].clone()
        if dref is None:
            dref = s
        derr = ((s - dref).abs().max() / dref.abs().max()).item()
        worst = max(worst, derr * 1e-2)   # (1e-12 on the checksum scale is 1e-10 here)
        assert derr < 1e-10, (step, derr)
    for k in opts:
        ctx.set_option(k, 0)
    assert err < 1e-12 and bad == 0, (step, opts, err, bad)
    if step % 50 == 49:
        print(f"step {step + 1}: worst relative checksum error so far {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {a.steps} steps of {a.workload}, worst relative checksum error {worst:.2e}")
