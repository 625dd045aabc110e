"""The ABI's size limit exercised for real (include/gridhip.h, "Limits"): one call of 2^31 - 256 visibilities."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_largest_call_the_abi_takes():
    """tools/max_size_check.py: n = 2^31 - 256 on the headline shape, device-resident (about 180 GB of HBM) - the grid's
    analytic checksum, errors == 0, nothing dropped, the tap-reusing path, degrid2's first and last 10^6 predictions
    against calls of those visibilities alone, and n + 1 refused.  No oracle grids 2 x 10^9 visibilities; the properties
    checked do not depend on the size."""
    import torch
    torch.cuda.empty_cache()  # (what earlier tests of this process left in torch's cache)
    free, _ = torch.cuda.mem_get_info(0)
    if free < 200 * 2**30:
        pytest.skip(f"needs ~180 GB of free HBM, {free / 2**30:.0f} GiB free")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "max_size_check.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "max size ok" in out.stdout


@pytest.mark.gpu
def test_two_host_threads_with_a_context_each():
    """A context is not thread-safe, but contexts are independent of each other (include/gridhip.h): two host threads,
    each with its own context and stream on the same GPU, grid and degrid concurrently (ctypes releases the GIL during
    the calls) - every grid's checksum and every prediction sample must come out as in a single-threaded run."""
    import threading

    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
    import bench
    import gridhip
    dev = torch.device("cuda:0")
    N, W, Q, S, n = 2048, 32, 8, 15, 3_000_000
    gcf = bench.synth_kernels(W, Q, S, dev)
    results, errors = {}, []

    def work(t):
        try:
            ctx = gridhip.Context(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                u, v, wb, vis = bench.synth_vis(n, N, W, S, 100 + t, dev)
                expect, scale = bench.expected_checksum(u, v, wb, vis, gcf, N)
                G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
                worst, first = 0.0, None
                for i in range(60):
                    if i % 7 == 3:
                        ctx.set_option("wgroups", (1, 2, 4, 8)[(i // 7) % 4])
                    G.zero_()
                    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
                    worst = max(worst, abs(G.sum().item() - expect.item()) / scale.item())
                    if i % 10 == 0:
                        d = ctx.degrid2(gcf, G, (u, v, None), wb)[::997].clone()
                        if first is None:
                            first = d
                        worst = max(worst, ((d - first).abs().max() / first.abs().max()).item())
                stream.synchronize()
                results[t] = (worst, ctx.get_option("errors"))
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors
    assert sorted(results) == [0, 1]
    for t, (worst, errs) in results.items():
        assert worst < 1e-10 and errs == 0, (t, worst, errs)
    del np
