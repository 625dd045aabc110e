#!/usr/bin/env python3
"""Would running call i+1's binning pre-pass beside call i's tile kernel pay?  (Round 1 tried it with that round's kernels
and it did not: profiles/r01_async_prepass.txt.)  Two contexts on one GPU: A grids through a plan (tile kernel only,
~10.4 ms at cfg3), B creates plans (pre-pass only, ~1.4 ms), B's i-th pre-pass ordered after A's (i-1)-th tile kernel so
that the two run as a pipelined library would run them.  Reported per option set of A (yield_cus lets B's work-groups
onto CUs while A's persistent kernel runs): A alone, B alone, both - against their sum and against A alone.
usage: python tools/prepass_overlap_probe.py [cfg3|cfg5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
import torch  # noqa: E402

import bench  # noqa: E402
import gridhip  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n, N, W, Q, S = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
A, B = gridhip.Context(0), gridhip.Context(0)
gcf = bench.synth_kernels(W, Q, S, dev)
u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
K = 8
with torch.cuda.stream(sA):
    planA = A.plan((N, N), gcf.shape, (u, v, None), wb)
    planA.grid(gcf, G, vis)
torch.cuda.synchronize()


def run(do_a, do_b):
    plans = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for i in range(K):
        if do_a:
            with torch.cuda.stream(sA):
                planA.grid(gcf, G, vis)
                e = torch.cuda.Event()
                e.record(sA)
                evs.append(e)
        if do_b:
            with torch.cuda.stream(sB):
                if do_a and i > 0:
                    sB.wait_event(evs[i - 1])  # pre-pass i runs beside tile kernel i, not earlier
                plans.append(B.plan((N, N), gcf.shape, (u, v, None), wb))  # (returns when its pre-pass is done)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    for p in plans:
        p.close()
    return dt


print(f"# {wl}: {n} vis, {N}^2; {K} iterations each; ms per iteration")
print("option_of_A     A_alone  B_alone  both   sum    both_minus_A_alone")
for opt in ("", "yield_cus=64", "yield_cus=128", "reserve_cus=32"):
    A.set_option("yield_cus", 0)
    A.set_option("reserve_cus", 0)
    if opt:
        k, val = opt.split("=")
        A.set_option(k, int(val))
    run(True, True)
    a, b, ab = run(True, False), run(False, True), run(True, True)
    print(f"{opt or 'default':14s}  {a:7.3f}  {b:7.3f}  {ab:6.3f}  {a + b:6.3f}  {ab - a:6.3f}", flush=True)
