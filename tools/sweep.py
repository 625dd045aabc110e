#!/usr/bin/env python3
"""Tuning sweep on one GPU: generate the bench workload once, time convgrid2 under several option sets.
usage: python tools/sweep.py [--workload cfg3] [--nvis N] [--dist uniform] "tile=64,block=1024" "tile=32,block=512" ...
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import numpy as np
import torch
import bench
import gridhip

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg3")
ap.add_argument("--nvis", type=int, default=0)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--wplanes", type=int, default=0)
ap.add_argument("--grid", type=int, default=0)
ap.add_argument("--support", type=int, default=0)
ap.add_argument("sets", nargs="*")
a = ap.parse_args()
n, N, W, Q, S = bench.WORKLOADS[a.workload]
if a.nvis:
    n = a.nvis
if a.wplanes:
    W = a.wplanes
if a.grid:
    N = a.grid
if a.support:
    S = a.support
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
gcf = bench.synth_kernels(W, Q, S, dev)
u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev, dist=a.dist)
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
ctx.enable_timing(True)
keys = ("tile", "tile_x", "tile_y", "block", "chunk", "wgroups", "variant", "sort", "dbg", "prepass", "coarse_shift", "scatter_chunk", "count_unroll", "wtable", "bigtile", "subfoot", "reserve_cus", "yield_cus")
for s in a.sets or [""]:
    for k in keys:
        try:
            ctx.set_option(k, 0)
        except gridhip.GridHipError:
            pass  # ("dbg" exists in the tuning build only: GRIDHIP_LIB=.../libgridhip_tuning.so)
    for kv in filter(None, s.split(",")):
        k, val = kv.split("=")
        ctx.set_option(k, int(val))
    try:
        for _ in range(1 if s is not (a.sets or [""])[0] else 6):  # (the process's first calls run ~7 % slow: clocks still rising)
            ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
        ts = []
        for _ in range(a.reps):
            ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
            ts.append(ctx.last_timing())
        t = np.array(ts).min(axis=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
        e1.record()
        torch.cuda.synchronize()
        per = e0.elapsed_time(e1) / 8
        print(f"{s or 'default':45s} total {t[0]:8.2f} ms  prepass {t[1]:7.2f}  kernel {t[2]:8.2f}  -> {n / t[0] / 1e3:8.1f} Mvis/s"
              f"   | 8 calls back to back: {per:6.2f} ms each -> {n / per / 1e3:8.1f} Mvis/s"
              f"   | {ctx.get_option('last_wgroups')} w-groups, tile {ctx.get_option('last_tile_y')} x {ctx.get_option('last_tile_x')}"
              f"{' (all of the LDS)' if ctx.get_option('last_bigtile') else ''}, path {ctx.get_option('last_path')}", flush=True)
    except Exception as e:
        print(f"{s:45s} FAILED {e}", flush=True)
