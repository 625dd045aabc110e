#!/usr/bin/env python3
"""Measure the secondary configurations on one GPU (the headline number comes from bench.py):
cfg2 grid, cfg3 grid with the 'core' distribution, cfg3 degrid, plans, cfg4 aw-gridding with and without the per-key
cache, the cfg5 share, a support sweep, do_imaging at N = 2400, the host-pointer ABI.  Prints one JSON line per measurement; timings are HIP-event totals (pre-pass + kernel) or wall time."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import numpy as np
import torch
import bench
import gridhip

dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
ctx.enable_timing(True)


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        fn()
        ts.append(ctx.last_timing())
    return np.array(ts).min(axis=0)


def wall(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def report(name, n, t_ms, extra=None):
    d = {"what": name, "n": n, "ms": round(float(t_ms), 3), "Mvis_per_s": round(n / float(t_ms) / 1e3, 1)}
    d.update(extra or {})
    print(json.dumps(d), flush=True)


which = sys.argv[1:] or ["cfg2", "core", "degrid", "aw", "cfg5", "plan", "supports", "imaging", "host"]
if "host" in which:
    # PCIe-inclusive rate of the drop-in host-pointer ABI (pageable numpy arrays in, grid out)
    n, N, W, Q, S = 20_000_000, 4096, 128, 8, 15
    gcf = bench.synth_kernels(W, Q, S, dev).cpu().numpy()
    u, v, wb, vis = (t.cpu().numpy() for t in bench.synth_vis(n, N, W, S, 9, dev))
    G = np.zeros((N, N), dtype=np.complex128)
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    t0 = time.perf_counter()
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    dt = (time.perf_counter() - t0) * 1e3
    report("cfg3 shape through the HOST-pointer ABI (H2D of 48 B/vis + 29.5 MB kernels + 256 MiB grid both ways)", n, dt)
    del gcf, u, v, wb, vis, G
if "plan" in which:
    n, N, W, Q, S = bench.WORKLOADS["cfg3"]
    gcf = bench.synth_kernels(W, Q, S, dev)
    u, v, wb, vis = bench.synth_vis(n, N, W, S, 6, dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    out = torch.empty(n, dtype=torch.complex128, device=dev)
    t = wall(lambda: ctx.plan((N, N), tuple(gcf.shape), (u, v, None), wb).close())
    report("cfg3 plan creation (binning only)", n, t)
    plan = ctx.plan((N, N), tuple(gcf.shape), (u, v, None), wb)
    report("cfg3 grid through a plan (no pre-pass)", n, wall(lambda: plan.grid(gcf, G, vis), 5))
    report("cfg3 degrid through a plan (no pre-pass)", n, wall(lambda: plan.degrid(gcf, G, out), 5))
    plan.close()
    del gcf, u, v, wb, vis, G, out
if "cfg2" in which:
    n, N, W, Q, S = bench.WORKLOADS["cfg2"]
    gcf = bench.synth_kernels(W, Q, S, dev)
    u, v, wb, vis = bench.synth_vis(n, N, W, S, 1, dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    t = timed(lambda: ctx.convgrid2(gcf, G, (u, v, None), wb, vis), 5)
    report("cfg2 grid: 1e6 vis, 2048^2, 7x7, 16 planes", n, t[0], {"prepass_ms": round(float(t[1]), 3), "kernel_ms": round(float(t[2]), 3)})
    del gcf, u, v, wb, vis, G
if "core" in which or "degrid" in which:
    n, N, W, Q, S = bench.WORKLOADS["cfg3"]
    gcf = bench.synth_kernels(W, Q, S, dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    if "core" in which:
        u, v, wb, vis = bench.synth_vis(n, N, W, S, 2, dev, dist="core")
        t = timed(lambda: ctx.convgrid2(gcf, G, (u, v, None), wb, vis))
        report("cfg3 grid, centrally concentrated uv (distribution B)", n, t[0], {"prepass_ms": round(float(t[1]), 3), "kernel_ms": round(float(t[2]), 3)})
        del u, v, wb, vis
    if "degrid" in which:
        u, v, wb, vis = bench.synth_vis(n, N, W, S, 3, dev)
        G.copy_(torch.complex(torch.randn((N, N), dtype=torch.float64, device=dev), torch.randn((N, N), dtype=torch.float64, device=dev)))
        out = torch.empty(n, dtype=torch.complex128, device=dev)
        t = timed(lambda: ctx.degrid2(gcf, G, (u, v, None), wb, out))
        report("cfg3 degrid2 (uniform)", n, t[0], {"prepass_ms": round(float(t[1]), 3), "kernel_ms": round(float(t[2]), 3)})
        del u, v, wb, vis, out
    del gcf, G
if "aw" in which:
    # config 4: aw-projection, 4096^2, 15x15, 128 planes, 512 antennas, per-antenna kernel lookup; baseline-structured
    # stream (bench.synth_aw_stream), with and without the per-key kernel cache
    N, W, Q, S, A = 4096, 128, 8, 15, 512
    gcf = bench.synth_kernels(W, Q, S, dev)
    ak = bench.synth_akernels(A, S, dev)
    for n in (1_000_000, 10_000_000):
        u, v, wb, a1, a2, vis = bench.synth_aw_stream(n, N, W, S, A, 5, dev)
        G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
        for cache in (1, 0):
            ctx.set_option("aw_cache", cache)
            t = wall(lambda: ctx.convgrid4(gcf, ak, G, (u, v, None), (wb, a1, a2), vis), 3)
            st = ctx.aw_stats(S)
            report(f"cfg4 aw-gridding (convgrid4), {A} antennas, per-key kernel cache {'on' if cache else 'off'}", n, t,
                   {"kernels_built": st["kernels_built"], "hit_rate": round(st["hit_rate"], 4),
                    "build_ms": round(st.get("build_ms", 0.0), 3), "grid_ms": round(st.get("grid_ms", 0.0), 3)})
        ctx.set_option("aw_cache", 1)
        del u, v, wb, a1, a2, vis, G
if "cfg5" in which:
    n, N, W, Q, S = bench.WORKLOADS["cfg5"]
    gcf = bench.synth_kernels(W, Q, S, dev)
    u, v, wb, vis = bench.synth_vis(n, N, W, S, 7, dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    t = timed(lambda: ctx.convgrid2(gcf, G, (u, v, None), wb, vis), 3)
    report("cfg5 share of one GPU: 1.25e8 vis, 8192^2, 15x15, 128 planes", n, t[0],
           {"prepass_ms": round(float(t[1]), 3), "kernel_ms": round(float(t[2]), 3)})
    del gcf, u, v, wb, vis, G
if "supports" in which:
    # cfg3's shape with other supports: squares through the tap-reusing kernel directly (5..16) or cut into square
    # parts (above 16), and a non-square one.  GRIDHIP_SWEEP_N sets the stream length (default 2e7; 1e8 = cfg3's own,
    # where a 31x31 item holds as many visibilities per distinct slice as the 15x15 headline case)
    n, N, W, Q = int(float(os.environ.get("GRIDHIP_SWEEP_N", "2e7"))), 4096, 128, 8
    u, v, wb, vis = bench.synth_vis(n, N, W, 31, 8, dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    for gh, gw in ((5, 5), (7, 7), (9, 9), (11, 11), (13, 13), (15, 15), (16, 16), (17, 17), (21, 21), (25, 25), (31, 31), (9, 5)):
        gcf = bench.synth_kernels(W, Q, max(gh, gw), dev)[..., :gh, :gw].contiguous()
        t = timed(lambda: ctx.convgrid2(gcf, G, (u, v, None), wb, vis), 3)
        report(f"support {gh}x{gw}", n, t[0], {"prepass_ms": round(float(t[1]), 3), "kernel_ms": round(float(t[2]), 3),
                                               "Gtaps_per_s_kernel": round(n * gh * gw / float(t[2]) / 1e6, 1)})
        del gcf
    del u, v, wb, vis, G
if "imaging" in which:
    # do_imaging (src/Gridding.hs:509-549) end to end at the driver's size, N = theta * lam = 2400
    # (src/ImageDataset.hs:32-33): mirror, weights, two gridding passes (image, PSF), Hermitian fill, two 2400^2
    # non-power-of-two hipFFTs, normalisation.  Host arrays in, images out (PCIe included).
    theta, lam = 0.008, 300000
    rng = np.random.default_rng(3)
    for n in (1_000_000, 10_000_000):
        uvw = np.stack([rng.uniform(-0.45 * lam, 0.45 * lam, n), rng.uniform(-0.45 * lam, 0.45 * lam, n),
                        rng.uniform(0, 20000, n)], axis=1)
        vis = rng.normal(size=n) + 1j * rng.normal(size=n)
        z = np.zeros(n, dtype=np.int64)
        for name, imgfn in (("simple_imaging", ("simple",)),
                            ("w_cache_imaging (wstep 2000, qpx 4, npixFF 256, 15x15)", ("w_cache", dict(wstep=2000, qpx=4, npixFF=256, npixKern=15)))):
            ctx.do_imaging(theta, lam, uvw, z, z, z, z, vis, imgfn)
            t0 = time.perf_counter()
            img, psf, pmax = ctx.do_imaging(theta, lam, uvw, z, z, z, z, vis, imgfn)
            dt = (time.perf_counter() - t0) * 1e3
            report(f"do_imaging N=2400, {name}, host arrays in / images out", n, dt, {"pmax": float(pmax)})
            # the resident form (gridhip_do_imaging_dev): uvw, vis, image and psf stay in HBM, nothing crosses PCIe
            duvw, dvis = torch.from_numpy(uvw).to(dev), torch.from_numpy(vis).to(dev)
            ts = []
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dimg, dpsf, dpmax = ctx.do_imaging(theta, lam, duvw, z, z, z, z, dvis, imgfn)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            report(f"do_imaging N=2400, {name}, device-resident (gridhip_do_imaging_dev)", n, min(ts[1:]),
                   {"pmax": float(dpmax), "first_call_ms": round(ts[0], 3), "calls_ms": [round(x, 3) for x in ts[1:]],
                    "same_as_host_form": bool(abs(dpmax - pmax) <= 1e-12 * abs(pmax))})
            del duvw, dvis, dimg, dpsf
        del uvw, vis
