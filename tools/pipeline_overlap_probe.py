#!/usr/bin/env python3
"""Does a collective-like kernel overlap the NEXT step's gridding when it is issued where bench.py issues it - on a
side stream, ordered (event) after the step's tile kernel - and does that need CUs reserved?

tools/reserve_cus_probe.py shows the worst case: a kernel with a collective's footprint queued in the MIDDLE of the
persistent tile kernel waits for its end unless 32 CUs are reserved.  In the pipeline the collective becomes ready at a
kernel boundary, at the same moment as the next step's pre-pass: it can take its CUs as the previous tile kernel's
work-groups retire, and the next tile kernel's persistent work-groups then start on what is left and pull the same
queues.  This probe runs that pipeline on one GPU with the stand-in of tools/micro/fat_copy.hip (64 work-groups, 197
VGPRs + 32 KB LDS each) repeated `passes` times per step to last about as long as a collective would, and reports per
(reserve_cus, passes): the step time against the pipeline without any side-stream work, and how long after its issue
the stand-in finished.   usage: python tools/pipeline_overlap_probe.py [cfg3|cfg5] [reserve_cus|yield_cus] [one-launch]"""
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
bufs = [torch.zeros((N, N), dtype=torch.complex128, device=dev) for _ in range(2)]
y0 = mirrored_first_row(N, S)
src = torch.zeros((N - y0, N), dtype=torch.complex128, device=dev)
dst = torch.empty_like(src)
side = torch.cuda.Stream()
fat = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libfatcopy.so"))
fat.fat_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
fat.fat_copy_rep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
ONE_LAUNCH = len(sys.argv) > 3 and sys.argv[3] == "one-launch"  # the stand-in as one long-lived kernel instead of `passes` launches
ev = lambda: torch.cuda.Event(enable_timing=True)

# how long one pass of the stand-in takes on an idle GPU
torch.cuda.synchronize()
with torch.cuda.stream(side):
    a, b = ev(), ev()
    fat.fat_copy(dst.data_ptr(), src.data_ptr(), src.numel(), 64, side.cuda_stream, None)
    a.record(side)
    for _ in range(10):
        fat.fat_copy(dst.data_ptr(), src.data_ptr(), src.numel(), 64, side.cuda_stream, None)
    b.record(side)
torch.cuda.synchronize()
one = a.elapsed_time(b) / 10
print(f"# {wl}: {n} vis, {N}^2 grid; stand-in: {src.numel() * 16 / 1e6:.0f} MB copied by 64 fat work-groups, {one:.3f} ms per pass on an idle GPU")
print("reserve_cus  passes  stand_in_alone_ms  step_ms_no_side_work  step_ms_with_stand_in  stand_in_issue_to_done_ms  hidden")
STEPS = 12
OPT = sys.argv[2] if len(sys.argv) > 2 else "reserve_cus"  # or "yield_cus"
print(f"# option {OPT}; stand-in as {'one launch' if ONE_LAUNCH else 'a train of launches'}")
for reserve in (0, 32) if OPT == "reserve_cus" else (32, 64):
    ctx.set_option(OPT, reserve)
    for passes in (0, 8, 24, 60):
        done = [None, None]
        spans = []
        torch.cuda.synchronize()
        t0, t1 = ev(), ev()
        for i in range(STEPS + 2):
            if i == 2:
                t0.record()
            g = bufs[i % 2]
            if done[i % 2] is not None:
                torch.cuda.current_stream().wait_event(done[i % 2][1])  # this buffer's previous "reduction"
            g.zero_()
            ctx.convgrid2(gcf, g, (u, v, None), wb, vis)
            if passes:
                e = ev()
                e.record()
                side.wait_event(e)
                s0, s1 = ev(), ev()
                s0.record(side)
                if ONE_LAUNCH:
                    fat.fat_copy_rep(dst.data_ptr(), src.data_ptr(), src.numel(), 64, passes, side.cuda_stream, None)
                else:
                    for _ in range(passes):
                        fat.fat_copy(dst.data_ptr(), src.data_ptr(), src.numel(), 64, side.cuda_stream, None)
                s1.record(side)
                done[i % 2] = (s0, s1)
                spans.append((s0, s1))
        t1.record()
        torch.cuda.synchronize()
        step = t0.elapsed_time(t1) / STEPS
        if passes == 0:
            base = step
            print(f"{reserve:11d}  {passes:6d}  {0.0:17.3f}  {step:20.3f}  {'':>21s}  {'':>25s}", flush=True)
        else:
            span = sorted(x.elapsed_time(y) for x, y in spans[2:])[len(spans[2:]) // 2]
            print(f"{reserve:11d}  {passes:6d}  {one * passes:17.3f}  {base:20.3f}  {step:21.3f}  {span:25.3f}  "
                  f"{'yes' if step < base + 0.5 * one * passes else 'no'}", flush=True)
ctx.set_option(OPT, 0)
