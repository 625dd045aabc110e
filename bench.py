#!/usr/bin/env python3
"""bench.py — Mvis/s of the w-projection gridder (convgrid2 semantics) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5]

One process per GPU.  Started under `python -m torch.distributed.run --nproc-per-node N` the ranks come from
the environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); started as a plain command with --gpus N > 1 this
process only spawns the N rank processes (before anything touches a GPU) and relays rank 0's JSON line.
Visibilities are sharded across ranks - ONE global, counter-based stream (visibility k's values depend on (seed, k)
only), of which rank r grids a contiguous range onto a private N x N complex128 grid - and the partial grids are
summed with ONE RCCL fp64 collective per step, issued on a side stream so that it overlaps the next step's gridding.
--scaling weak (default): the stream has n_gpus x vis_per_gpu visibilities, every GPU grids vis_per_gpu of them;
--scaling strong: the stream has the workload's visibilities in total, every GPU grids 1 / n_gpus of them (the
N-GPU grid then equals the 1-GPU grid of the same stream).  After the timed steps every run certifies itself: the sum
of the (reduced) grid is compared with the analytic checksum sum_k vis_k * sum_ij K[slice_k] of the stream (all taps are
in range by construction) and the process exits non-zero if they differ by more than 1e-10 or the library counted an
internal error.

A "step" = binning pre-pass + tile kernel over the rank's whole shard (+ grid clear and all-reduce when N > 1),
inputs already resident in HBM.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  cfg3 (default; configs[2], the configuration the metric is quoted on): 10^8 synthetic visibilities, 4096^2 grid,
       128 w-planes, 15x15 support, oversampling Q = 8, uniform uv distribution (SURVEY.md §8d distribution A)
  cfg2 (configs[1]): 10^6 vis, 2048^2, 16 planes, 7x7
  cfg5 (configs[4], per-GPU share): 1.25 x 10^8 vis per GPU, 8192^2 grid, 128 planes, 15x15
  cfg4 (configs[3]): aw-projection (convgrid4), 10^6 vis, 4096^2 grid, 15x15, 128 planes, 512 antennas
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LDS_ATOMIC_B_PER_CLK = 64.0  # one 64-lane ds_add_f64 (512 B of operands) per 8 LDS cycles per CU
#                              (tools/micro/lds_atomic.hip, profiles/r01_lds_atomic_microbench.txt)
FP64_VALU_FLOP_PER_CLK_CU = 128.0  # 4 SIMDs x 16 lanes x FMA (78.6 TFLOP/s vector fp64 at 2.4 GHz)

WORKLOADS = {
    #         n            N     W    Q  S
    "cfg3": (100_000_000, 4096, 128, 8, 15),
    "cfg2": (1_000_000, 2048, 16, 8, 7),
    "cfg5": (125_000_000, 8192, 128, 8, 15),
    "cfg4": (1_000_000, 4096, 128, 8, 15),
}
AW_ANTENNAS = 512


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--nvis", type=int, default=0, help="override visibilities per GPU")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "core"])
    ap.add_argument("--cpu-sample", type=int, default=0, help="visibilities of the CPU baseline's sample (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU grids the workload's visibilities (the stream grows with N); strong: the GPUs "
                         "share the workload's visibilities (n / N each)")
    ap.add_argument("--collective", default="torch", choices=["torch", "cabi", "cabi-rs"],
                    help="N > 1: the per-step sum of the partial grids, on a side stream beside the next step's gridding: "
                         "torch = torch.distributed all_reduce (nccl = RCCL); cabi = libgridhip's own communicator "
                         "(gridhip_comm_*, ncclAllReduce); cabi-rs = the same with ncclReduceScatter + ncclAllGather")
    ap.add_argument("--reduce-rows", default="auto", choices=["auto", "all"],
                    help="auto: the stream is mirrored (v >= 0), so only the rows it can touch are reduced (half the bytes)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "side", "inline"],
                    help="N > 1: where the per-step collective runs: side = on a side stream beside the next step's gridding; "
                         "inline = right behind the step's gridding, no overlap; auto = an untimed pass runs a few steps "
                         "in line and on the side stream (plain, with 64 CUs yielding, with 32 CUs reserved) and the "
                         "fastest schedule - max over ranks - is used (tools/pipeline_overlap_probe.py)")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="compute units the persistent tile kernel leaves idle for a side-stream collective's kernel (-1 = "
                         "measured, see --overlap; 32 = one per shader engine of every XCD, profiles/r03_reserve_cus.txt)")
    ap.add_argument("--yield-cus", type=int, default=-1,
                    help="compute units on which the tile kernel runs short-lived work-groups instead of persistent ones, so "
                         "that a kernel queued on another stream gets them within a few hundred microseconds (-1 = "
                         "measured, see --overlap; multiples of 32)")
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x5EEDC0DE)
    ap.add_argument("--grids", type=int, default=0,
                    help="N = 1: how many zeroed grids the steps rotate over (0 = one per step, at most 16; 1 = every step "
                         "accumulates into the same grid, as the round-2 bench did)")
    ap.add_argument("--opt", action="append", default=[], help="gridhip option key=value (tile, block, chunk, wgroups, variant, sort)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per tile-kernel launch from a separate rocprofv3 --pmc pass "
                         "(default: profiles/traffic.json if it was measured on this kernel source)")
    ap.add_argument("--aw-cache", type=int, default=1, help="cfg4: per-key aw-kernel de-duplication (1 = on)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only (gloo, no GPU): checks the rank spawning / environment plumbing on a CPU box")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# plain `python bench.py --gpus N`: spawn the ranks.  Nothing here may touch a GPU (no torch.cuda call, no HIP).
def spawn_ranks(args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(len(procs)))
        while pending:
            for i in sorted(pending):
                code = procs[i].poll()
                if code is None:
                    continue
                pending.discard(i)
                if code != 0:
                    rc = rc or code
                    for j in pending:  # one rank failed: the others would wait in a collective for ever
                        procs[j].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ---------------------------------------------------------------------------------------------------------
def alg_bytes_per_vis(S):
    """40 B stream + per tap 16 B kernel read + 16 B grid read + 16 B grid write."""
    return 40 + 48 * S * S


def compulsory_bytes_per_vis(n, N, W, Q, S):
    """SURVEY.md §8(d) B_min: stream once, grid read + written once, kernel table read once."""
    return 40 + (32.0 * N * N + 16.0 * W * Q * Q * S * S) / n


def lds_atomic_cycles_per_vis(S):
    """LDS cycles the accumulate loop of the tap-reusing tile kernel needs per visibility: two ds_add_f64 (re, im)
    per step of 64 taps, 8 cycles per 64-lane instruction, 7 with three 16-lane groups active, 6 with two or fewer
    (measured: tools/micro/lds_atomic.hip, profiles/r01_lds_atomic_microbench.txt).  Mirrors the kernel's own
    constants (tile_sorted.hip: TAIL0, EXTRA, NSTEP_ALL, TAIL, PAIR); supports above 16 x 16 take their steps in
    parts of at most four, which does not change the count."""
    taps = S * S
    tail0 = taps - ((taps + 63) // 64 - 1) * 64
    # a last step of 33 or 34 taps keeps 32 (two visibilities of a run then share it: one full-width instruction pair
    # for both); a last step of one or two taps (supports above 16) disappears; the taps left over go once per block
    # of 64 records, every lane for its own record
    extra = tail0 - 32 if 32 < tail0 <= 34 else tail0 if (taps > 256 and tail0 <= 2) else 0
    covered = taps - extra
    nstep = (covered + 63) // 64
    tail = covered - (nstep - 1) * 64
    last = 2 * 8 / 2 if (tail == 32 and extra) else 2 * (6 if tail <= 32 else 7 if tail <= 48 else 8)
    return (nstep - 1) * 16 + last + extra * 2 * 8 / 64


def csrc_fingerprint():
    """sha256 over the sources of the measured kernels (pre-pass + tile kernels), comments and white space left out:
    a PMC measurement is only quoted for the code it was taken on."""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.startswith(("tile_", "bin", "common")) and name.endswith((".hip", ".h")):
            src = open(os.path.join(d, name), "r", encoding="utf-8", errors="replace").read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            src = re.sub(r"//[^\n]*", "", src)
            h.update(name.encode())
            h.update(re.sub(r"\s+", "", src).encode())
    return h.hexdigest()[:16]


def committed_traffic(workload):
    """HBM bytes per tile-kernel launch from profiles/traffic.json (rocprofv3 --pmc passes, tools/profile.sh), or
    None when that file was measured on other kernel sources than the ones now in the tree."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[workload]
    except (OSError, KeyError, ValueError):
        return None, "no PMC measurement committed for this workload"
    if rec.get("csrc_sha16") != csrc_fingerprint():
        return None, "profiles/traffic.json was measured on other kernel sources (csrc_sha16 differs): not quoted"
    return rec["hbm_bytes_per_launch"], f"profiles/traffic.json ({rec.get('source', 'rocprofv3 --pmc')})"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def stats(xs):
    import numpy as np
    a = np.asarray(xs, dtype=np.float64)
    return {"mean": float(a.mean()), "median": float(np.median(a)), "min": float(a.min()), "max": float(a.max())}


# ---------------------------------------------------------------------------------------------------------
def synth_kernels(W, Q, S, device):
    """Deterministic smooth complex kernels exp(-r^2/sigma^2) * exp(i*phi(w, r))."""
    import torch
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


def synth_akernels(A, S, device):
    """Deterministic per-antenna kernels [A,S,S]: a narrow Gaussian with an antenna-dependent phase ramp."""
    import torch
    j = torch.arange(S, dtype=torch.float64, device=device) - S // 2
    a = torch.arange(A, dtype=torch.float64, device=device)
    r2 = j[:, None] ** 2 + j[None, :] ** 2
    amp = torch.exp(-r2 / 4.0)
    phase = 0.01 * (a[:, None, None] + 1.0) * (j[None, :, None] - 0.5 * j[None, None, :])
    return torch.polar(amp[None].expand(A, S, S).contiguous(), phase).contiguous()


# Counter-based stream: every value of visibility k is a function of (seed, k, j) only - splitmix64 of
# seed + (8 k + j) * golden, j = the draw's number - so any rank can draw any range of the one global stream, on any
# device, and the CPU twin (tests/test_bench_helpers.py: numpy uint64) produces the same coordinates bit for bit
# (integer arithmetic, then exact conversions and single IEEE operations).
_GOLD, _C1, _C2 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB
_DRAWS = 8  # counters per visibility: 0 u, 1 v, 2 w, 3 / 4 the value's Box-Muller pair, 5 / 6 "core" distribution, 7 spare


def _s64(c):
    """a 64-bit constant as the signed value torch's int64 arithmetic (two's complement, wrapping) holds"""
    c &= (1 << 64) - 1
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)  # logical shift of an int64 tensor (>> is arithmetic)


def counter_uniform(seed, k, j):
    """U[0, 1) with 53 random bits for visibility numbers k (int64 tensor) and draw j."""
    import torch
    z = (k * _DRAWS + j) * _s64(_GOLD) + _s64(seed)
    z = (z ^ _lsr(z, 30)) * _s64(_C1)
    z = (z ^ _lsr(z, 27)) * _s64(_C2)
    z = z ^ _lsr(z, 31)
    return _lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def counter_uniform_numpy(seed, k, j):
    """The CPU twin of counter_uniform in numpy uint64 arithmetic (k: integer array)."""
    import numpy as np
    m = (1 << 64) - 1
    with np.errstate(over="ignore"):
        z = (np.asarray(k).astype(np.uint64) * np.uint64(_DRAWS) + np.uint64(j)) * np.uint64(_GOLD) + np.uint64(seed & m)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synth_vis(n, N, W, S, seed, device, wstep=2000, dist="uniform", lo=0):
    """Visibilities [lo, lo + n) of the global synthetic stream, generated on `device` in slabs."""
    import math
    import torch
    u = torch.empty(n, dtype=torch.float64, device=device)
    v = torch.empty(n, dtype=torch.float64, device=device)
    wb = torch.empty(n, dtype=torch.int64, device=device)
    vis = torch.empty(n, dtype=torch.complex128, device=device)
    m = (S / 2 + 1) / N  # margin so that every tap is in range
    slab = 1 << 24

    def normal_pair(k, j):  # Box-Muller on draws j, j + 1 (1 - U is in (0, 1]: the logarithm is finite)
        r = torch.sqrt(-2.0 * torch.log(1.0 - counter_uniform(seed, k, j)))
        t = (2.0 * math.pi) * counter_uniform(seed, k, j + 1)
        return r * torch.cos(t), r * torch.sin(t)

    for a in range(0, n, slab):
        b = min(n, a + slab)
        k = torch.arange(lo + a, lo + b, dtype=torch.int64, device=device)
        if dist == "uniform":
            pu = (counter_uniform(seed, k, 0) - 0.5) * (1 - 2 * m)
            pv = (counter_uniform(seed, k, 1) - 0.5) * (1 - 2 * m)
        else:  # "core": centrally concentrated, contention stress
            gu, gv = normal_pair(k, 5)
            pu = (gu * 0.08).clamp(-0.5 + m, 0.5 - m)
            pv = (gv * 0.08).clamp(-0.5 + m, 0.5 - m)
        # mirror_uvw (src/Gridding.hs:558-561): v >= 0
        neg = pv < 0
        pu = torch.where(neg, -pu, pu)
        pv = torch.where(neg, -pv, pv)
        ww = counter_uniform(seed, k, 2) * (W * wstep)
        # w-bin rule of w_cache_imaging (src/Gridding.hs:426-432), clamped to the planes we have
        u[a:b], v[a:b] = pu, pv
        wb[a:b] = torch.round(ww / wstep).to(torch.int64).clamp_(0, W - 1)
        re, im = normal_pair(k, 3)
        vis[a:b] = torch.complex(re, im)
    return u, v, wb, vis


def expected_checksum(u, v, wb, vis, gcf, N):
    """sum(G) a correct convgrid2 of this stream produces when every tap is in range: sum_k vis_k * sum_ij K[slice_k],
    with the slice index recomputed from frac_coord's formula (src/Gridding.hs:126-140) in torch fp64; and the
    absolute-value scale sum_k |vis_k| * sum_ij |K[slice_k]| the difference is judged against.  Slabs: no
    stream-sized temporaries survive."""
    import torch
    W, Q = gcf.shape[0], gcf.shape[1]
    ksum = gcf.sum(dim=(3, 4)).reshape(-1)
    kabs = gcf.abs().sum(dim=(3, 4)).reshape(-1)
    tot = torch.zeros((), dtype=torch.complex128, device=u.device)
    scale = torch.zeros((), dtype=torch.float64, device=u.device)

    def frac(p):
        x = N // 2 + p * N
        fl = torch.floor(x + 0.5 / Q)
        return torch.round((x - fl) * Q).clamp_(0, Q - 1).to(torch.int64)

    slab = 1 << 24
    for a in range(0, u.shape[0], slab):
        b = min(u.shape[0], a + slab)
        idx = (wb[a:b] * Q + frac(v[a:b])) * Q + frac(u[a:b])
        tot += (vis[a:b] * ksum[idx]).sum()
        scale += (vis[a:b].abs() * kabs[idx]).sum()
    return tot, scale


def synth_aw_stream(n, N, W, S, A, seed, device, dumps=8, drift_cells=0.02, wstep=2000):
    """Baseline-structured stream for the aw workload: n / dumps random baselines (a1 < a2), each observed for `dumps`
    consecutive samples during which its uv point drifts by `drift_cells` grid cells per sample in a fixed random
    direction (what time / frequency sampling does to a baseline's track; real SKA1-Low spacings drift far less per
    sample).  Consecutive samples therefore mostly share (a1, a2, wbin, yf, xf) - the repetition the per-key kernel
    cache exploits; the measured hit rate is reported.  src/ImageDataset.hs:88-104 reads one antenna pair per
    visibility."""
    import math
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed ^ 0xA11CE)
    nb = (n + dumps - 1) // dumps
    rnd = lambda k: torch.rand(k, generator=gen, device=device, dtype=torch.float64)
    m = (S / 2 + 1 + dumps * drift_cells) / N
    pu0 = (rnd(nb) - 0.5) * (1 - 2 * m)
    pv0 = rnd(nb) * (0.5 - m)  # mirrored: v >= 0
    ang = rnd(nb) * (2 * math.pi)
    a1 = (rnd(nb) * (A - 1)).to(torch.int64)
    a2 = (a1 + 1 + (rnd(nb) * (A - 1 - a1).to(torch.float64)).to(torch.int64)).clamp_(max=A - 1)
    wb0 = torch.round(rnd(nb) * W).to(torch.int64).clamp_(0, W - 1)
    d = torch.arange(dumps, device=device, dtype=torch.float64)
    rep = lambda t: t.repeat_interleave(dumps)[:n].contiguous()
    dd = d.repeat(nb)[:n]
    u = rep(pu0) + dd * rep(torch.cos(ang)) * (drift_cells / N)
    v = rep(pv0) + dd * rep(torch.sin(ang)).abs() * (drift_cells / N)
    re = torch.randn(n, generator=gen, device=device, dtype=torch.float64)
    im = torch.randn(n, generator=gen, device=device, dtype=torch.float64)
    return u.contiguous(), v.contiguous(), rep(wb0), rep(a1), rep(a2), torch.complex(re, im)


# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(ctx, u, v, wb, vis, gcf, N, n, sample_override):
    """Time the CPU oracle (C/OpenMP restatement of src/Gridding.hs:199-244, NOT Accelerate) on a bounded sample,
    all three of its threading modes; `value` is the best of them.  The oracle's grid of the largest sample is then
    also the checker of the GPU's grid of the same sample, cell by cell (`parity_rel_err`)."""
    import numpy as np
    from oracle import gridref_c
    gridref_c.build()
    hk = gcf.cpu().numpy()
    cores = gridref_c.max_threads()
    names = {0: "shared grid + atomic updates (what a parallel permute (+) does)", 1: "private grids + reduction",
             2: "owner computes: bands of grid rows, one thread each"}
    modes = {}
    parity = None
    for mode in (2, 1, 0):
        sample = sample_override or (40_000_000 if mode == 2 else 4_000_000)
        sample = min(sample, n)
        # (private grids: 16 threads at most - every thread zeroes and reduces its own N x N grid)
        threads = min(cores, 16) if mode == 1 else cores
        hu, hv = u[:sample].cpu().numpy(), v[:sample].cpu().numpy()
        hw, hvis = wb[:sample].cpu().numpy(), vis[:sample].cpu().numpy()
        G = np.zeros((N, N), dtype=np.complex128)
        t0 = time.perf_counter()
        gridref_c.convgrid2(hk, G, hu, hv, hw, hvis, mt_mode=mode, nthreads=threads)
        dt = time.perf_counter() - t0
        modes[names[mode]] = {"Mvis_per_s": sample / dt / 1e6, "sample_vis": sample, "seconds": dt, "threads": threads}
        if mode == 2:  # the GPU's grid of the same visibilities against it
            import torch
            Gd = torch.zeros((N, N), dtype=torch.complex128, device=u.device)
            ctx.convgrid2(gcf, Gd, (u[:sample], v[:sample], None), wb[:sample], vis[:sample])
            ref = torch.from_numpy(G).to(u.device)
            parity = {"rel_err": float(((Gd - ref).abs().max() / ref.abs().max()).item()), "sample_vis": sample}
            del Gd, ref
    best = max(modes, key=lambda k: modes[k]["Mvis_per_s"])
    return {
        "value": modes[best]["Mvis_per_s"],
        "unit": "Mvis/s",
        "cores": modes[best]["threads"],
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": f"first {modes[best]['sample_vis']} visibilities of the same workload, C/OpenMP oracle "
                  f"(oracle/gridref.c, restatement of src/Gridding.hs:199-244; the Accelerate llvm-native path cannot be "
                  f"built), mode '{best}', {modes[best]['seconds']:.2f} s",
        "modes": modes,
        "parity_rel_err": parity["rel_err"] if parity else None,
        "parity_note": (f"max |G_gpu - G_oracle| / max |G_oracle| over all cells, both gridding the first {parity['sample_vis']} "
                        "visibilities of the workload (tolerance 1e-10)") if parity else None,
    }


def cpu_baseline_aw(ctx, u, v, wb, a1, a2, vis, wk, ak, N, sample):
    import numpy as np
    import torch
    from oracle import gridref_c
    gridref_c.build()
    f = lambda t: t[:sample].cpu().numpy()
    G = np.zeros((N, N), dtype=np.complex128)
    t0 = time.perf_counter()
    gridref_c.awgrid(wk.cpu().numpy(), ak.cpu().numpy(), G, f(u), f(v), f(wb), f(a1), f(a2), f(vis))
    dt = time.perf_counter() - t0
    g = lambda t: t[:sample]
    Gd = torch.zeros((N, N), dtype=torch.complex128, device=u.device)
    ctx.convgrid4(wk, ak, Gd, (g(u), g(v), None), (g(wb), g(a1), g(a2)), g(vis))
    ref = torch.from_numpy(G).to(u.device)
    perr = float(((Gd - ref).abs().max() / ref.abs().max()).item())
    return {"value": sample / dt / 1e6, "unit": "Mvis/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
            "parity_rel_err": perr,
            "parity_note": f"max |G_gpu - G_oracle| / max |G_oracle|, both gridding the first {sample} visibilities (tolerance 1e-10)",
            "sample": f"first {sample} visibilities of the same workload, C oracle of convgrid4 (oracle/gridref.c, "
                      f"FFT-free restatement of src/Gridding.hs:318-396,761-811), single thread, {dt:.2f} s"}


# ---------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": float(t.item()),
                              "workload": args.workload, "vis_per_gpu": WORKLOADS[args.workload][0]}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    # (test hooks: GRIDHIP_BENCH_SHARE_GPU=1 puts every rank on device 0 and GRIDHIP_BENCH_BACKEND=gloo replaces RCCL,
    # which refuses two ranks on one device, by gloo's host-staged collectives - how a one-GPU box runs this file with
    # two real ranks, tests/test_gpu_distributed.py)
    if os.environ.get("GRIDHIP_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:  # (counting devices does not initialise the GPU)
        raise SystemExit(f"rank {rank}: needs GPU {local_rank}, this machine has {torch.cuda.device_count()}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # (GRIDHIP_BENCH_FORCE_DIST=1: run the N > 1 code path - process group, reducer, all-reduce - with one rank,
    # which is how a one-GPU box exercises it)
    if world > 1 or os.environ.get("GRIDHIP_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("GRIDHIP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import gridhip
    from gridhip.distributed import (Comm, InlineGridReducer, OverlappedCommReducer, OverlappedGridReducer,
                                     mirrored_first_row, shard_bounds)
    n, N, W, Q, S = WORKLOADS[args.workload]
    if args.nvis:
        n = args.nvis
    aw = args.workload == "cfg4"
    # the one global stream and this rank's range of it
    if args.scaling == "strong":
        n_total = n
        lo, hi = shard_bounds(n_total, world, rank)
    else:
        n_total = n * world
        lo, hi = rank * n, (rank + 1) * n
    n_rank = hi - lo
    n = n_rank  # visibilities per launch on this rank: what the per-kernel figures below are per
    ctx = gridhip.Context(local_rank)
    for kv in args.opt:
        k, val = kv.split("=")
        ctx.set_option(k, int(val))
    if aw:
        ctx.set_option("aw_cache", args.aw_cache)
    reserve = max(args.reserve_cus, 0)  # (N > 1: decided below, together with where the collective runs)
    yield_cus = max(args.yield_cus, 0)
    if yield_cus:
        ctx.set_option("yield_cus", yield_cus)
    gcf = synth_kernels(W, Q, S, device)
    if aw:
        akerns = synth_akernels(AW_ANTENNAS, S, device)
        u, v, wb, a1, a2, vis = synth_aw_stream(n_rank, N, W, S, AW_ANTENNAS, args.seed + rank, device)
    else:
        u, v, wb, vis = synth_vis(n_rank, N, W, S, args.seed, device, dist=args.dist, lo=lo)
    # N = 1: every step grids onto a grid that was zeroed BEFORE the timed region ("grid re-zeroed outside
    # the timed region") - a ring of zeroed grids, one per step (more steps than grids: the ring wraps and a grid
    # receives several whole passes, which the check below accounts for).
    # N > 1: two buffers used alternately; a step clears its buffer (each step reduces its own partial grids).
    nbuf = 2 if dist is not None else max(1, min(args.grids or (args.steps + args.warmup), 16))
    bufs = [torch.zeros((N, N), dtype=torch.complex128, device=device) for _ in range(nbuf)]
    passes = [0] * nbuf
    G = bufs[0]
    # N > 1: one fp64 sum of the partial grids per step over xGMI (RCCL), issued on a side stream so that it overlaps
    # the next step's gridding.  The stream is mirrored (v >= 0): rows the footprints cannot reach stay exactly zero
    # and are left out of the collective.
    rows = (mirrored_first_row(N, S), N) if (args.reduce_rows == "auto" and not aw) else None
    red = comm = None
    schedule = None
    if dist is not None:
        if args.collective != "torch":
            comm = Comm.from_torch(ctx)
            comm.set_option("collective", 1 if args.collective == "cabi-rs" else 0)
        part0 = bufs[0] if rows is None else bufs[0][rows[0]:rows[1]]

        def collective_once():
            if comm is None:
                dist.all_reduce(torch.view_as_real(part0))
            elif rows is None:
                comm.allreduce_grid(bufs[0])
            else:
                comm.allreduce_grid_rows(bufs[0], *rows)

        # Where the collective runs is MEASURED, not assumed (tools/pipeline_overlap_probe.py, reserve_cus_probe.py and
        # profiles/r03_pipeline_overlap.txt, r03_yield_cus.txt show why no rule is safe): a collective's kernel issued on a
        # side stream behind the step's tile kernel takes its CUs at the kernel boundary, as that kernel's work-groups
        # retire, and runs beside the next step's gridding - but a SECOND kernel of the same collective (reduce-scatter
        # then all-gather) becomes ready in the middle of a persistent tile kernel that occupies every CU and then
        # waits for its end, unless CUs come free: "yield_cus" (64 CUs run short-lived work-groups, +0.7 % on the
        # gridding) or "reserve_cus" (32 CUs left idle, +10 %).  A short collective can be cheapest in line.  auto: an
        # untimed pass runs a few steps of each schedule; the slowest rank's time decides for all ranks.
        ctx.enable_timing(True)
        ctx._use_torch_stream()

        def make_reducer(side):
            if side and comm is None:
                return OverlappedGridReducer(bufs, rows=rows)
            if side:
                return OverlappedCommReducer(comm, bufs, rows=rows)
            return InlineGridReducer(bufs, rows=rows, comm=comm)

        def trial(side, res, yld, k=6):
            ctx.set_option("reserve_cus", res)
            ctx.set_option("yield_cus", yld)
            r = make_reducer(side)
            t_ = 0.0
            for i in range(2 + k):
                if i == 2:
                    r.finish()
                    torch.cuda.synchronize()
                    dist.barrier()
                    t_ = time.perf_counter()
                g_ = r.begin(i)
                if aw:
                    ctx.convgrid4(gcf, akerns, g_, (u, v, None), (wb, a1, a2), vis)
                else:
                    ctx.convgrid2(gcf, g_, (u, v, None), wb, vis)
                r.end(i)
            r.finish()
            torch.cuda.synchronize()
            dist.barrier()
            t_ = (time.perf_counter() - t_) / k * 1e3
            r.close()
            return t_

        if args.overlap == "inline":
            cands = [(False, 0, 0)]
        elif args.reserve_cus >= 0 or args.yield_cus >= 0:
            cands = [(True, max(args.reserve_cus, 0), max(args.yield_cus, 0))]
        else:
            cands = [(True, 0, 0), (True, 0, 64), (True, 32, 0)]
        if args.overlap == "auto":
            cands.insert(0, (False, 0, 0))
        trial(*cands[0], k=1)  # first use of everything: plans, RCCL's own warm-up
        ms = torch.tensor([trial(*c_) for c_ in cands], dtype=torch.float64, device=device) if len(cands) > 1 else torch.zeros(1, dtype=torch.float64, device=device)
        dist.all_reduce(ms, op=dist.ReduceOp.MAX)
        best = int(torch.argmin(ms).item())
        side, reserve, yield_cus = cands[best]
        names = [("side stream" if c_[0] else "in line") + (f", {c_[1]} CUs reserved" if c_[1] else "") +
                 (f", {c_[2]} CUs yielding" if c_[2] else "") for c_ in cands]
        schedule = {"collective_runs": "side stream, beside the next step's gridding" if side else "in line, behind the step's gridding",
                    "chosen_by": "measurement" if len(cands) > 1 else "flags",
                    "tried_ms_per_step": {nm: round(float(x), 4) for nm, x in zip(names, ms.tolist())} if len(cands) > 1 else None,
                    "rule": "the fastest of the tried schedules (max over ranks of an untimed 6-step trial each)"}
        red = make_reducer(side)
        ctx.set_option("yield_cus", yield_cus)
        for b_ in bufs:
            b_.zero_()
    ctx.set_option("reserve_cus", reserve)
    counter = [0]

    def step():
        i = counter[0]
        counter[0] += 1
        g = red.begin(i) if red else bufs[i % nbuf]
        if red:
            passes[i % 2] = 0
        passes[i % nbuf] += 1
        if aw:
            ctx.convgrid4(gcf, akerns, g, (u, v, None), (wb, a1, a2), vis)
        else:
            ctx.convgrid2(gcf, g, (u, v, None), wb, vis)  # this rank's range of the stream
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
    ctx.enable_timing(True)  # restart the event ring: the timed steps are calls 0 .. K-1
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if red:
        red.finish()  # every step's all-reduce is complete inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-step device times of the timed steps, from HIP events recorded on the kernels' own stream
    kept = min(args.steps, 64)
    times = [ctx.timing(back) for back in range(kept)]
    ker_ms, pre_ms = [t[2] for t in times], [t[1] for t in times]
    clock_ghz = ctx.get_option("clock_khz") / 1e6
    errors = ctx.get_option("errors")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([float(errors)], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        errors = int(t.item())

    # ---- the run certifies itself (outside the timed region): the sum of every grid the steps produced against the
    # analytic checksum of the stream.  N > 1: the reduced grids hold the contributions of ALL ranks (rows the
    # collective skipped must then be zero on every rank, or the sums disagree), so the expected value is all-reduced.
    check = None
    if not aw:
        tc0 = time.perf_counter()
        expect, scale = expected_checksum(u, v, wb, vis, gcf, N)
        ex = torch.stack([expect.real, expect.imag, scale])
        if dist is not None:
            dist.all_reduce(ex)
        expect, scale = complex(ex[0].item(), ex[1].item()), ex[2].item()
        worst, nonzero = 0.0, 0
        for b, g in enumerate(bufs):
            if passes[b] == 0:
                continue
            got = complex(g.sum().item())
            worst = max(worst, abs(got - passes[b] * expect) / (passes[b] * scale))
            nonzero = max(nonzero, int((g != 0).sum().item()))
        if dist is not None:
            t = torch.tensor([worst], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst = float(t.item())
        torch.cuda.synchronize()
        check = {"rel_err": worst, "tolerance": 1e-10, "cells_nonzero": nonzero, "grids_checked": sum(1 for x in passes if x),
                 "what": "max over the grids the timed and warm-up steps produced of |sum(G) - passes * sum_k vis_k * "
                         "sum_ij K[slice_k]| / (passes * sum_k |vis_k| * sum_ij |K[slice_k]|); slices recomputed from "
                         "frac_coord's formula in torch; N > 1: G is the reduced grid, the expectation is summed over ranks",
                 "seconds": time.perf_counter() - tc0}

    # N > 1: what the collective costs on its own (outside the timed region; gridding and all-reduce separately)
    multi = None
    if dist is not None:
        part = G if rows is None else G[rows[0]:rows[1]]
        tg = torch.view_as_real(part)

        def one():
            if args.collective == "torch":
                dist.all_reduce(tg)
            elif rows is None:
                comm.allreduce_grid(G)
            else:
                comm.allreduce_grid_rows(G, *rows)

        red.close()  # (the communicator's collectives go back to the context's stream = torch's current stream)
        ctx._use_torch_stream()
        for _ in range(2):
            one()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            one()
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 5
        nbytes = int(part.numel() * 16)
        multi = {"rccl_ranks": dist.get_world_size(), "collective": args.collective, "scaling": args.scaling,
                 "reduced_rows": list(rows) if rows else [0, N], "reserve_cus": reserve, "yield_cus": yield_cus, "schedule": schedule,
                 "allreduce_bytes": nbytes, "allreduce_ms_alone": ar_ms,
                 "allreduce_busbw_GBps": nbytes * 2 * (world - 1) / world / (ar_ms * 1e-3) / 1e9 if world > 1 else None,
                 "gridding_ms_per_step": float(np.mean(ker_ms) + np.mean(pre_ms)),
                 "combined_ms_per_step": elapsed / args.steps * 1e3,
                 "check_rel_err": check["rel_err"] if check else None}

    failed = []
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_vis = n_total
        k_avg = float(np.mean(ker_ms))
        props = torch.cuda.get_device_properties(device)
        cus = props.multi_processor_count
        nominal_ghz = getattr(props, "clock_rate", 2_400_000) / 1e6
        opts = {k: ctx.get_option(k) for k in ("tile", "block", "chunk", "wgroups", "variant", "sort", "prepass")}
        if aw:
            # dominant kernel: the per-key aw-kernel build (fp64 vector ALU): (2S-1)^2-bounded 'same' convolutions
            info = ctx.aw_stats(S)
            flops = info["conv_flops_per_call"]
            if n > (1 << 22):  # the library works in batches of 2^22 visibilities and its events time the first one
                flops *= (1 << 22) / n
            peak = cus * FP64_VALU_FLOP_PER_CLK_CU * nominal_ghz / 1e3  # TFLOP/s
            build_ms = float(np.mean(pre_ms))  # pair kernels + keys + aw_build_kernel (the last dominates)
            achieved = flops / (build_ms * 1e-3) / 1e12 if build_ms > 0 else 0.0
            roof = {"bound": "valu_f64", "kernel": f"gridhip::aw_build_kernel<{S}> (one 'same' convolution of the pair kernel "
                                                   "with the w-kernel slice per distinct (a1, a2, wbin, yf, xf))",
                    "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                    "what": "fp64 flops of the kernel builds (8 per complex product, 28 561 products per 15x15 kernel) over the "
                            "device time of the build phase; peak = CUs x 128 flop/clk x the nominal clock",
                    "flops_per_launch": flops, "kernel_ms_avg": build_ms, "build_ms": stats(pre_ms),
                    "grid_ms": stats(ker_ms), "aw_cache": args.aw_cache, "aw": info,
                    "clock_GHz_held_by_the_builder": ctx.get_option("aw_clock_khz") / 1e6, "clock_GHz_nominal": nominal_ghz}
            metric = "Mvis/s gridded (aw-proj, 4096^2 grid)"
            what = f"aw-projection grid (convgrid4): {n_rank} vis/GPU, {N}^2 grid, {W} w-planes, {AW_ANTENNAS} antennas, {S}x{S} support, Q={Q}"
        else:
            clk = clock_ghz if clock_ghz > 0.5 else nominal_ghz
            lds_bytes = 2.0 * S * S * 8.0 * n  # operand bytes of the LDS atomics one launch needs (re + im per tap)
            achieved = lds_bytes / (k_avg * 1e-3) / 1e9
            peak = cus * LDS_ATOMIC_B_PER_CLK * clk  # GB/s at the clock the kernel held
            floor_ms = n * lds_atomic_cycles_per_vis(S) / cus / (clk * 1e9) * 1e3
            traffic, traffic_src = (args.traffic_bytes, "--traffic-bytes") if args.traffic_bytes else (None, None)
            if traffic is None and not args.opt and not args.nvis and args.dist == "uniform" and n == WORKLOADS[args.workload][0] and not reserve:
                traffic, traffic_src = committed_traffic(args.workload)
            bmin = compulsory_bytes_per_vis(n, N, W, Q, S)
            alg = alg_bytes_per_vis(S) * n / (k_avg * 1e-3) / 1e9
            hbm = {"alg_bytes_per_vis": alg_bytes_per_vis(S), "alg_GBps": alg, "alg_over_hbm_peak": alg / HBM_PEAK_GBPS,
                   "note": "SURVEY §8(d)'s algorithmic figure charges 32 B/tap of grid read-modify-write and 16 B/tap of kernel "
                           "reads to HBM; the tile design keeps the RMW in LDS and the taps in L2, so alg_over_hbm_peak is "
                           "not a fraction of anything physical - the bounded figures are roofline.frac (LDS atomic unit) "
                           "and hbm_measured_frac",
                   "compulsory_bytes_per_vis": bmin, "traffic_source": traffic_src}
            if traffic:
                hbm.update(measured_GBps=traffic / (k_avg * 1e-3) / 1e9,
                           hbm_measured_frac=traffic / (k_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                           traffic_over_compulsory=traffic / (bmin * n))
            roof = {
                "bound": "lds_atomic",
                "kernel": f"gridhip::tile_grid_sorted_kernel<{S},false> (tile_grid_kernel when sort is off)",
                "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                "traffic": traffic,
                "what": "LDS-atomic operand bytes per second: 2 x S^2 fp64 ds_add_f64 lane-operations per visibility; "
                        "peak = CUs x 64 B/clk (one 64-lane ds_add_f64 per 8 LDS cycles) x the shader clock measured "
                        "inside the kernel",
                "clock_GHz": clk, "clock_source": "s_memtime / s_memrealtime stamps inside the tile kernel" if clock_ghz > 0.5 else "nominal",
                "cus": cus, "lds_cycles_per_vis": lds_atomic_cycles_per_vis(S), "lds_floor_ms": floor_ms,
                "lds_floor_frac": floor_ms / k_avg,
                "kernel_ms_avg": k_avg, "kernel_ms": stats(ker_ms), "prepass_ms_avg": float(np.mean(pre_ms)),
                "prepass_ms": stats(pre_ms), "kernel_Mvis_per_s": n / (k_avg * 1e-3) / 1e6,
                "hbm": hbm,
            }
            metric = "Mvis/s gridded (w-proj, 4096^2 grid)" if N == 4096 else f"Mvis/s gridded (w-proj, {N}^2 grid)"
            what = f"w-projection grid (convgrid2): {n_rank} vis/GPU, {N}^2 grid, {W} w-planes, {S}x{S} support, Q={Q}, {args.dist} uv"
        for key in ("frac", "lds_floor_frac"):
            if key in roof:
                assert 0.0 < roof[key] <= 1.0, f"roofline.{key} = {roof[key]} is not a fraction"
        out = {
            "metric": metric,
            "value": total_vis / (elapsed / args.steps) / 1e6,
            "unit": "Mvis/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "step_ms_device": stats([a + b for a, b in zip(ker_ms, pre_ms)]),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": what,
                "name": args.workload,
                "vis_per_gpu": n_rank, "vis_total": n_total, "scaling": args.scaling, "seed": args.seed, "grid": N, "w_planes": W, "support": S, "oversample": Q,
                "parallelism": f"vis-sharded x{world}" + (f" + RCCL fp64 grid sum per step ({args.collective}; "
                                                          f"{schedule['collective_runs']}; {reserve} CUs reserved, {yield_cus} yielding)"
                                                          if world > 1 else ""),
                "scaling_note": ("weak: one global counter-based stream of n_gpus x vis_per_gpu visibilities, rank r grids "
                                 "[r, r + 1) x vis_per_gpu of it; value = vis_total / step time" if args.scaling == "weak" else
                                 "strong: one global counter-based stream of vis_total visibilities, rank r grids its 1 / n_gpus "
                                 "share; value = vis_total / step time"),
                "options": opts,
            },
            "errors": errors,
            "check": check,
            "roofline": roof,
        }
        if multi:
            out["multi_gpu"] = multi
        if world == 1 and not args.no_cpu:
            if aw:
                out["cpu_baseline"] = cpu_baseline_aw(ctx, u, v, wb, a1, a2, vis, gcf, akerns, N, min(args.cpu_sample or 20_000, n))
            else:
                out["cpu_baseline"] = cpu_baseline(ctx, u, v, wb, vis, gcf, N, n, args.cpu_sample)
            pe = out["cpu_baseline"].get("parity_rel_err")
            if pe is not None and not pe <= 1e-10:
                failed.append(f"GPU grid of the CPU baseline's sample differs from the oracle's: {pe:.3e}")
        print(json.dumps(out), flush=True)
    if errors:
        failed.append(f"the library counted {errors} internal consistency errors")
    if check is not None and not check["rel_err"] <= check["tolerance"]:
        failed.append(f"checksum of the gridded stream is off by {check['rel_err']:.3e} (relative)")
    if comm:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        print("bench.py: RESULT CHECK FAILED: " + "; ".join(failed), file=sys.stderr, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
