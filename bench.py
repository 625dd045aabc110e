#!/usr/bin/env python3
"""bench.py — Mvis/s of the w-projection gridder (convgrid2 semantics) on MI355X.

One process per GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`),
visibilities sharded across ranks (weak scaling: every rank grids its own `--nvis` shard onto a
private N x N complex128 grid), one RCCL fp64 all-reduce of the partial grids per step.

A "step" = binning pre-pass + tile kernel over the rank's whole shard (+ all-reduce when N > 1),
inputs already resident in HBM.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  10^8 synthetic visibilities, 4096^2 grid, 128 w-planes, 15x15 support, oversampling Q = 8,
  uniform uv distribution (SURVEY.md §8d distribution A).
`--workload cfg2` selects configs[1] (10^6 vis, 2048^2, 7x7).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    #         n            N     W    Q  S
    "cfg3": (100_000_000, 4096, 128, 8, 15),
    "cfg2": (1_000_000, 2048, 16, 8, 7),
}


def alg_bytes_per_vis(S):
    """40 B stream + per tap 16 B kernel read + 16 B grid read + 16 B grid write."""
    return 40 + 48 * S * S


def lds_atomic_cycles_per_vis(S):
    """LDS cycles the accumulate loop needs per visibility: two ds_add_f64 (re, im) per step of 64 taps, 8 cycles
    per 64-lane instruction, 7 with three 16-lane groups active, 6 with two or fewer (measured:
    tools/micro/lds_atomic.hip, profiles/r01_lds_atomic_microbench.txt)."""
    taps = S * S
    full, tail = divmod(taps, 64)
    cyc = full * 2 * 8
    if 32 < tail <= 34:  # the last step keeps 32 taps and serves two visibilities of a run at once (a lone one
        # pays 2 x 6 cycles for it); the other taps go once per block of 64 records
        cyc += 2 * 8 / 2 + (tail - 32) * 2 * 8 / 64
    elif tail:
        cyc += 2 * (6 if tail <= 32 else 7 if tail <= 48 else 8)
    return cyc


def synth_kernels(W, Q, S, device):
    """Deterministic smooth complex kernels exp(-r^2/sigma^2) * exp(i*phi(w, r))."""
    j = torch.arange(S, dtype=torch.float64, device=device) - S // 2
    q = torch.arange(Q, dtype=torch.float64, device=device) / Q
    w = torch.arange(W, dtype=torch.float64, device=device)
    yy = (j[None, None, :, None] - q[:, None, None, None])  # [Q,1,S,1]
    xx = (j[None, None, None, :] - q[None, :, None, None])  # [1,Q,1,S]
    r2 = yy * yy + xx * xx                                  # [Q,Q,S,S]
    sigma2 = (S / 3.0) ** 2
    amp = torch.exp(-r2 / sigma2)
    phase = 0.02 * (w[:, None, None, None, None] + 1.0) * r2[None]
    return torch.polar(amp[None].expand(W, Q, Q, S, S).contiguous(), phase).contiguous()


def synth_vis(n, N, W, S, seed, device, wstep=2000, dist="uniform"):
    """Counter-free seeded synthetic stream, generated on the device in slabs."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    u = torch.empty(n, dtype=torch.float64, device=device)
    v = torch.empty(n, dtype=torch.float64, device=device)
    wb = torch.empty(n, dtype=torch.int64, device=device)
    vis = torch.empty(n, dtype=torch.complex128, device=device)
    m = (S / 2 + 1) / N  # margin so that every tap is in range
    slab = 1 << 24
    for lo in range(0, n, slab):
        hi = min(n, lo + slab)
        k = hi - lo
        if dist == "uniform":
            pu = (torch.rand(k, generator=gen, device=device, dtype=torch.float64) - 0.5) * (1 - 2 * m)
            pv = (torch.rand(k, generator=gen, device=device, dtype=torch.float64) - 0.5) * (1 - 2 * m)
        else:  # "core": centrally concentrated, contention stress
            pu = (torch.randn(k, generator=gen, device=device, dtype=torch.float64) * 0.08).clamp(-0.5 + m, 0.5 - m)
            pv = (torch.randn(k, generator=gen, device=device, dtype=torch.float64) * 0.08).clamp(-0.5 + m, 0.5 - m)
        # mirror_uvw (src/Gridding.hs:558-561): v >= 0
        neg = pv < 0
        pu = torch.where(neg, -pu, pu)
        pv = torch.where(neg, -pv, pv)
        ww = torch.rand(k, generator=gen, device=device, dtype=torch.float64) * (W * wstep)
        # w-bin rule of w_cache_imaging (src/Gridding.hs:426-432), clamped to the planes we have
        b = torch.round(ww / wstep).to(torch.int64).clamp_(0, W - 1)
        re = torch.randn(k, generator=gen, device=device, dtype=torch.float64)
        im = torch.randn(k, generator=gen, device=device, dtype=torch.float64)
        u[lo:hi], v[lo:hi], wb[lo:hi] = pu, pv, b
        vis[lo:hi] = torch.complex(re, im)
    return u, v, wb, vis


def cpu_baseline(u, v, wb, vis, gcf, N, sample):
    """Time the CPU oracle (port of src/Gridding.hs, NOT Accelerate) on a bounded sample."""
    from oracle import gridref_c
    gridref_c.build()
    hu, hv = u[:sample].cpu().numpy(), v[:sample].cpu().numpy()
    hw, hvis = wb[:sample].cpu().numpy(), vis[:sample].cpu().numpy()
    hk = gcf.cpu().numpy()
    cores = gridref_c.max_threads()
    best, mode_best = None, None
    for mode in (1, 0):  # private grids + reduce, shared grid + atomics
        G = np.zeros((N, N), dtype=np.complex128)
        t0 = time.perf_counter()
        gridref_c.convgrid2(hk, G, hu, hv, hw, hvis, mt_mode=mode, nthreads=cores)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, mode_best = dt, mode
    return {
        "value": sample / best / 1e6,
        "unit": "Mvis/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {sample} visibilities of the same workload, C/OpenMP oracle "
                  f"({'private grids + reduce' if mode_best == 1 else 'shared grid + atomics'}), {best:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--nvis", type=int, default=0, help="override visibilities per GPU")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "core"])
    ap.add_argument("--cpu-sample", type=int, default=4_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="gridhip option key=value (tile, block, chunk, wgroups, variant, sort)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per tile-kernel launch from a separate rocprofv3 --pmc pass "
                         "(default: the committed measurement in profiles/traffic.json for this workload)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    import gridhip
    n, N, W, Q, S = WORKLOADS[args.workload]
    if args.nvis:
        n = args.nvis
    ctx = gridhip.Context(local_rank)
    for kv in args.opt:
        k, val = kv.split("=")
        ctx.set_option(k, int(val))

    gcf = synth_kernels(W, Q, S, device)
    u, v, wb, vis = synth_vis(n, N, W, S, 0x5EEDC0DE + rank, device, dist=args.dist)
    G = torch.zeros((N, N), dtype=torch.complex128, device=device)
    from gridhip.distributed import OverlappedGridReducer
    # N > 1: one fp64 sum all-reduce of the partial grids per step over xGMI (RCCL), issued on a side
    # stream so that it overlaps the next step's gridding (two grid buffers used alternately)
    red = OverlappedGridReducer([G, torch.zeros_like(G)]) if dist is not None else None
    counter = [0]

    def step():
        i = counter[0]
        counter[0] += 1
        g = red.begin(i) if red else G
        # this rank's shard of the stream (generated per rank: shard r of a world*n stream)
        ctx.convgrid2(gcf, g, (u, v, None), wb, vis)
        if red:
            red.end(i)

    ctx.enable_timing(True)
    for _ in range(args.warmup):
        step()
    if red:
        red.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ker_ms, pre_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        _, p, k = ctx.last_timing()  # HIP events on the kernels' own stream
        ker_ms.append(k)
        pre_ms.append(p)
    if red:
        red.finish()  # every step's all-reduce is complete inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    traffic = args.traffic_bytes
    if traffic is None and not args.opt and not args.nvis and args.dist == "uniform":
        # PMC counters cannot be collected inside this run; quote the committed rocprofv3 --pmc
        # measurement of the same workload (tools/profile.sh -> profiles/traffic.json)
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[args.workload]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_vis = n * world
        k_avg = float(np.mean(ker_ms))
        props = torch.cuda.get_device_properties(device)
        clock_ghz = getattr(props, "clock_rate", 2_400_000) / 1e6
        floor_ms = n * lds_atomic_cycles_per_vis(S) / props.multi_processor_count / (clock_ghz * 1e9) * 1e3
        lds_floor = {"cycles_per_vis": lds_atomic_cycles_per_vis(S), "cus": props.multi_processor_count,
                     "clock_GHz": clock_ghz, "floor_ms": floor_ms, "frac": floor_ms / k_avg}
        achieved = alg_bytes_per_vis(S) * n / (k_avg * 1e-3) / 1e9
        out = {
            "metric": "Mvis/s gridded (w-proj, 4096^2 grid)" if args.workload == "cfg3" else "Mvis/s gridded (w-proj)",
            "value": total_vis / (elapsed / args.steps) / 1e6,
            "unit": "Mvis/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"w-projection grid (convgrid2): {n} vis/GPU, {N}^2 grid, {W} w-planes, {S}x{S} support, Q={Q}, {args.dist} uv",
                "vis_per_gpu": n, "grid": N, "w_planes": W, "support": S, "oversample": Q,
                "parallelism": f"vis-sharded x{world}" + (" + RCCL fp64 grid all-reduce per step (overlapped with the next step's gridding)" if world > 1 else ""),
                "options": {k: ctx.get_option(k) for k in ("tile", "block", "chunk", "wgroups", "variant", "sort")},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "tile_grid_sorted_kernel<15> (tile_grid_kernel when sort is off)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "alg_bytes_per_vis": alg_bytes_per_vis(S),
                "kernel_ms_avg": k_avg,
                "prepass_ms_avg": float(np.mean(pre_ms)),
                "kernel_Mvis_per_s": n / (k_avg * 1e-3) / 1e6,
                # what actually binds the kernel: the LDS atomic unit, one per CU
                "lds_atomic": lds_floor,
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(u, v, wb, vis, gcf, N, min(args.cpu_sample, n))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
