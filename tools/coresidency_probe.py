#!/usr/bin/env python3
"""Does a memory-bound kernel on a second (non-blocking) stream run beside the persistent tile kernel, and what
does it cost the tile kernel?  (Feasibility probe for overlapping the next call's pre-pass.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch, bench, gridhip
n, N, W, Q, S = bench.WORKLOADS["cfg3"]
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
gcf = bench.synth_kernels(W, Q, S, dev)
u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
plan = ctx.plan((N, N), tuple(gcf.shape), (u, v, None), wb)
a = torch.empty(1 << 28, dtype=torch.float64, device=dev)  # 2 GiB
b = torch.empty_like(a)
s2 = torch.cuda.Stream()
ctx.enable_timing(True)

def tile():
    if plan is not None:
        plan.grid(gcf, G, vis)
    else:
        ctx.convgrid2(gcf, G, (u, v, None), wb, vis)

for _ in range(2):
    tile()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# standalone copy (4 GiB moved)
with torch.cuda.stream(s2):
    e0.record(); b.copy_(a); b.copy_(a); e1.record()
torch.cuda.synchronize()
print(f"copy x2 alone: {e0.elapsed_time(e1):.2f} ms")
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record(); tile(); t1.record(); torch.cuda.synchronize()
print(f"tile kernel alone: {t0.elapsed_time(t1):.2f} ms")
for ncopy in (2, 3, 4, 6):
    for rep in range(2):
        t0.record()
        tile()
        t1.record()
        with torch.cuda.stream(s2):
            e0.record()
            for _ in range(ncopy):
                b.copy_(a)
            e1.record()
        torch.cuda.synchronize()
        print(f"together: tile {t0.elapsed_time(t1):.2f} ms, copy x{ncopy} ({ncopy * 4.3:.1f} GB moved) {e0.elapsed_time(e1):.2f} ms")
