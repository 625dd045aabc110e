#!/usr/bin/env python3
"""What does the "reserve_cus" option cost the tile kernel on one GPU, and does a kernel queued on another stream
really start beside the persistent tile kernel once CUs are left free?

The stand-in for the collective is a device-to-device copy of the rows a mirrored 4096^2 grid reduces (135 MB read +
135 MB written; an RCCL all-reduce kernel is, like it, a few work-groups that stream memory), enqueued on a side stream
behind a spin kernel that holds it back until the tile kernel is 3 ms into its run (a plan: the pass is the tile
kernel alone, no pre-pass the copy could slip in beside).  Reported per k: the pass alone, the pass with the copy
beside it, and when the copy finished relative to the pass's start - "copy_end < pass_end" means it ran beside the
tile kernel instead of after it.  Two stand-ins: torch's copy kernel (few registers, no LDS: it fits into what the
tile kernel leaves free on the CUs it occupies) and tools/micro/fat_copy.hip (~110 VGPRs per lane and 32 KB of LDS
per work-group, as a collective's kernel has: it only fits on a CU the tile kernel does not occupy).
usage: python tools/reserve_cus_probe.py [cfg3|cfg5] [reserve_cus|yield_cus]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch  # noqa: E402

import bench  # noqa: E402
import gridhip  # noqa: E402
from gridhip.distributed import mirrored_first_row  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n, N, W, Q, S = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
ctx = gridhip.Context(0)
gcf = bench.synth_kernels(W, Q, S, dev)
u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
y0 = mirrored_first_row(N, S)
src = torch.zeros((N - y0, N), dtype=torch.complex128, device=dev)
dst = torch.empty_like(src)
side = torch.cuda.Stream()
ev = lambda: torch.cuda.Event(enable_timing=True)
fat = None
_so = os.path.join(ROOT, "tools", "micro", "libfatcopy.so")
if os.path.exists(_so):
    fat = ctypes.CDLL(_so)
    fat.fat_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
NFAT = 64
stamps = torch.zeros(NFAT * 4, dtype=torch.int64, device=dev)


plan = ctx.plan((N, N), tuple(gcf.shape), (u, v, None), wb)


def step():
    plan.grid(gcf, G, vis)


# spin kernel: how many cycles are 3 ms?
torch.cuda.synchronize()
a0, a1 = ev(), ev()
a0.record()
torch.cuda._sleep(10_000_000)
a1.record()
torch.cuda.synchronize()
SPIN = int(10_000_000 * 3.0 / a0.elapsed_time(a1))


print(f"# {wl}: {n} vis, {N}^2 grid, {S}x{S}; stand-in collective: copy of rows [{y0}, {N}) = {src.numel() * 16 / 1e6:.0f} MB")
with torch.cuda.stream(side):
    a, b = ev(), ev()
    dst.copy_(src)
    a.record()
    dst.copy_(src)
    b.record()
torch.cuda.synchronize()
print(f"copy alone: {a.elapsed_time(b):.3f} ms")
def copy_on_side(kind):
    if kind == "torch":
        dst.copy_(src)
    else:  # 64 work-groups of 256 threads
        rc = fat.fat_copy(dst.data_ptr(), src.data_ptr(), src.numel(), NFAT, side.cuda_stream, stamps.data_ptr())
        assert rc == 0


print("stand_in  cus_given_up  pass_alone_ms  pass_with_copy_ms  copy_start_ms  copy_end_ms  copy_ran_beside")
OPT = sys.argv[2] if len(sys.argv) > 2 else "reserve_cus"  # or "yield_cus"
for kind, k in [(kind, k) for kind in (("torch", "fat") if fat else ("torch",)) for k in (0, 8, 16, 24, 32, 64)]:
    ctx.set_option(OPT, k)
    ctx.enable_timing(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    alone = []
    for _ in range(5):
        t0, t1 = ev(), ev()
        t0.record()
        step()
        t1.record()
        torch.cuda.synchronize()
        alone.append(t0.elapsed_time(t1))
    both, cbeg, cend = [], [], []
    for _ in range(5):
        t0, t1, c0, c1 = ev(), ev(), ev(), ev()
        t0.record()
        side.wait_event(t0)
        step()
        t1.record()
        with torch.cuda.stream(side):
            torch.cuda._sleep(SPIN)  # the "collective" is issued 3 ms into the tile kernel
            c0.record()
            copy_on_side(kind)
            c1.record()
        torch.cuda.synchronize()
        both.append(t0.elapsed_time(t1))
        cbeg.append(t0.elapsed_time(c0))
        cend.append(t0.elapsed_time(c1))
    med = lambda x: sorted(x)[len(x) // 2]
    extra = ""
    if kind == "fat":  # the last run's work-groups: when each started (ms after the first one), and on how many CUs
        st = stamps.view(NFAT, 4).cpu()
        t0s = (st[:, 0] - st[:, 0].min()).double() / 1e5
        cus = len({(int(a), int(b) & 0xff00) for a, b in zip(st[:, 2].tolist(), st[:, 3].tolist())})  # (xcc, se/cu bits of HW_ID)
        extra = (f"  | work-groups started within 1 ms of the first: {int((t0s < 1.0).sum())} of {NFAT}; last start "
                 f"{t0s.max().item():.2f} ms after the first; on {cus} distinct (XCC, CU)")
    print(f"{kind:8s}  {k:11d}  {med(alone):13.3f}  {med(both):17.3f}  {med(cbeg):13.3f}  {med(cend):11.3f}  {'yes' if med(cend) < med(both) - 0.5 else 'no'}" + extra,
          flush=True)
ctx.set_option(OPT, 0)
