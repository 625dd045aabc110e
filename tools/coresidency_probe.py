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

# 3. hot-spot atomics beside the tile kernel: 24 M atomic adds onto 512 addresses (what the coarse scatter's
#    per-chunk reservations amount to), then onto 33 800 addresses (the fine scatter's)
for naddr in (512, 33800):
    idx = torch.randint(0, naddr, (24_000_000,), device=dev)
    ones = torch.ones(24_000_000, dtype=torch.int32, device=dev)
    acc = torch.zeros(naddr, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        e0.record(); acc.index_add_(0, idx, ones); e1.record()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1)
    for rep in range(2):
        t0.record(); tile(); t1.record()
        with torch.cuda.stream(s2):
            e0.record()
            for _ in range(2):
                acc.index_add_(0, idx, ones)
            e1.record()
        torch.cuda.synchronize()
        print(f"24M atomics on {naddr} addresses x2: alone {alone:.2f} ms each; together: tile {t0.elapsed_time(t1):.2f} ms, atomics {e0.elapsed_time(e1):.2f} ms")

# 4. scattered short runs beside the tile kernel: 1.6 GB written as randomly placed rows of 64 / 128 / 256 / 1024 B
#    (what the coarse level of the scatter writes with 2 048- and 4 096-record chunks, and longer ones)
src = torch.empty(1 << 27, dtype=torch.float64, device=dev)      # 1 GiB
dst = torch.empty(1 << 28, dtype=torch.float64, device=dev)      # 2 GiB
for row_bytes in (64, 128, 256, 1024):
    w = row_bytes // 8
    nrows = (1 << 27) // w
    perm = torch.randperm((1 << 28) // w, device=dev)[:nrows]
    s_rows, d_rows = src.view(nrows, w), dst.view(-1, w)
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        d_rows.index_copy_(0, perm, s_rows)
        e0.record(); d_rows.index_copy_(0, perm, s_rows); e1.record()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1)
    for rep in range(2):
        t0.record(); tile(); t1.record()
        with torch.cuda.stream(s2):
            e0.record()
            for _ in range(2):
                d_rows.index_copy_(0, perm, s_rows)
            e1.record()
        torch.cuda.synchronize()
        print(f"1 GiB as random {row_bytes}-B rows x2: alone {alone:.2f} ms each; together: tile {t0.elapsed_time(t1):.2f} ms, writes {e0.elapsed_time(e1):.2f} ms")
