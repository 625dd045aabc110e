#!/usr/bin/env python3
"""convgrid2 on a baseline-structured stream (bench.synth_aw_stream: `dumps` consecutive samples per baseline drifting
0.02 cell each - the order real visibilities arrive in) against the uniformly random stream of the headline, same
shape and count: what the stream's order is worth (runs of equal kernel slice, value gathers that share sectors).
usage: python tools/tracks_probe.py [dumps ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch  # noqa: E402

import bench  # noqa: E402
import gridhip  # noqa: E402

n, N, W, Q, S = bench.WORKLOADS["cfg3"]
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
ctx.enable_timing(True)
gcf = bench.synth_kernels(W, Q, S, dev)
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)


def run(tag, u, v, wb, vis):
    for _ in range(4):
        ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    ts = []
    for _ in range(5):
        ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
        ts.append(ctx.last_timing())
    t = min(ts)
    G.zero_()
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    expect, scale = bench.expected_checksum(u, v, wb, vis, gcf, N)
    rel = abs(G.sum().item() - expect.item()) / scale.item()
    print(f"{tag:42s} {t[0]:7.2f} ms (pre-pass {t[1]:.2f} + tile kernel {t[2]:.2f}) = {n / t[0] / 1e3:7.0f} Mvis/s; checksum {rel:.1e}", flush=True)


u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
run("uniformly random stream (the headline)", u, v, wb, vis)
del u, v, wb, vis
for dumps in [int(a) for a in sys.argv[1:]] or [8, 64]:
    u, v, wb, a1, a2, vis = bench.synth_aw_stream(n, N, W, S, bench.AW_ANTENNAS, 0x5EEDC0DE, dev, dumps=dumps)
    del a1, a2
    run(f"baseline tracks, {dumps} samples per baseline", u, v, wb, vis)
    del u, v, wb, vis
