#!/usr/bin/env python3
"""Is the device path capturable into a HIP graph?  After a warm-up call (scratch sized, LDS limits raised) a
gridhip_convgrid2_dev call enqueues only kernels and memsets on the caller's stream, so the caller can capture it
(here through torch.cuda.CUDAGraph on torch's capture stream) and replay it: one graph launch instead of ~9 kernel /
memset launches.  Prints per-call times eager vs replayed for cfg2 (launch-bound) and cfg3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch
import bench
import gridhip

dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
for wl in sys.argv[1:] or ["cfg2", "cfg3"]:
    n, N, W, Q, S = bench.WORKLOADS[wl]
    gcf = bench.synth_kernels(W, Q, S, dev)
    u, v, wb, vis = bench.synth_vis(n, N, W, S, 11, dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    ref = torch.zeros_like(G)
    ctx.convgrid2(gcf, ref, (u, v, None), wb, vis)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.convgrid2(gcf, G, (u, v, None), wb, vis)  # warm-up on the capture stream
    torch.cuda.synchronize()
    G.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    torch.cuda.synchronize()
    G.zero_()
    errs = []
    for _ in range(3):  # every replay must reset and refill the pre-pass tables: same grid each time
        G.zero_()
        g.replay()
        torch.cuda.synchronize()
        errs.append(((G - ref).abs().max() / ref.abs().max()).item())
    err = max(errs)
    print(f"{wl}: replays vs eager: rel err {errs}, errors {ctx.get_option('errors')}", flush=True)
    if err > 1e-10:
        print("  graph replay does not reproduce the eager result: not timing it")
        continue

    def timeit(fn, reps=50):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    reps = 200 if n <= 10**6 else 10
    t_eager = timeit(lambda: ctx.convgrid2(gcf, G, (u, v, None), wb, vis), reps)
    t_graph = timeit(lambda: g.replay(), reps)
    print(f"{wl}: replayed grid vs eager rel err {err:.2e}; per call eager {t_eager:.4f} ms -> {n / t_eager / 1e3:.0f} Mvis/s, "
          f"graph replay {t_graph:.4f} ms -> {n / t_graph / 1e3:.0f} Mvis/s, errors {ctx.get_option('errors')}", flush=True)
    del gcf, u, v, wb, vis, G, ref
