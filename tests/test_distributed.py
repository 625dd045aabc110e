"""N > 1 path on CPU: world_size-2 gloo processes shard the visibilities, grid their shard (the
CPU oracle stands in for the GPU gridder here — the sharding/all-reduce logic is what is under
test) and all-reduce the partial grids; the result must equal the single-process grid."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_shard_bounds_tile_the_stream():
    from gridhip.distributed import shard_bounds
    for n in (0, 1, 7, 8, 100, 10**8 + 3):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
    import torch.distributed as dist
    from gridhip.distributed import sharded_convgrid2
    from oracle import gridref_c
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)  # same stream on every rank
    N, W, Q, S, n = 96, 4, 4, 7, 5001
    gcf = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
    u, v = rng.uniform(-0.55, 0.55, n), rng.uniform(-0.55, 0.55, n)
    wb = rng.integers(0, W, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    G = np.zeros((N, N), dtype=np.complex128)
    gridder = lambda k, a, p, wbin, vv: gridref_c.convgrid2(k, a, p[0], p[1], wbin, vv)  # reference-style (gcf a p wbin v)
    sharded_convgrid2(gridder, gcf, G, (u, v, None), wb, vis, rank, world)
    ref = gridref_c.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis)
    # async form used by the overlapped reducer: the handle completes the same sum
    from gridhip.distributed import allreduce_grid
    H = np.full((4, 4), complex(rank + 1, -1.0))
    allreduce_grid(H, async_op=True).wait()
    assert np.allclose(H, complex(sum(range(1, world + 1)), -world))
    q.put((rank, float(np.abs(G - ref).max() / np.abs(ref).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_grid_matches_single_process():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "worker failed"
    res = [q.get(timeout=10) for _ in range(world)]
    assert sorted(r for r, _ in res) == [0, 1]
    for _, err in res:
        assert err < 1e-12  # same tolerance class as the GPU parity (summation order differs)


def _bench_worker(rank, world, port, scaling, q):
    """bench.py's N > 1 bookkeeping on two gloo ranks, the C oracle standing in for the GPU gridder: one global
    counter-based stream, rank r grids its range, only the rows a mirrored stream can touch are all-reduced, and
    every rank certifies the reduced grid against the all-reduced analytic checksum."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
    import torch
    import torch.distributed as dist
    import bench
    from gridhip.distributed import allreduce_grid, mirrored_first_row, shard_bounds
    from oracle import gridref_c
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, W, Q, S, n = 192, 8, 4, 7, 6001
    dev = torch.device("cpu")
    if scaling == "strong":
        n_total = n
        lo, hi = shard_bounds(n_total, world, rank)
    else:
        n_total = n * world
        lo, hi = rank * n, (rank + 1) * n
    gcf = bench.synth_kernels(W, Q, S, dev)
    u, v, wb, vis = bench.synth_vis(hi - lo, N, W, S, 0x5EEDC0DE, dev, lo=lo)
    G = np.zeros((N, N), dtype=np.complex128)
    gridref_c.convgrid2(gcf.numpy(), G, u.numpy(), v.numpy(), wb.numpy(), vis.numpy())
    rows = (mirrored_first_row(N, S), N)
    allreduce_grid(G, rows=rows)
    expect, scale = bench.expected_checksum(u, v, wb, vis, gcf, N)
    ex = torch.stack([expect.real, expect.imag, scale])
    dist.all_reduce(ex)
    rel = abs(G.sum() - complex(ex[0].item(), ex[1].item())) / ex[2].item()
    # and against the one-process grid of the whole stream
    U, V, WB, VIS = bench.synth_vis(n_total, N, W, S, 0x5EEDC0DE, dev)
    ref = gridref_c.convgrid2(gcf.numpy(), np.zeros((N, N), dtype=np.complex128), U.numpy(), V.numpy(), WB.numpy(), VIS.numpy())
    q.put((rank, float(rel), float(np.abs(G - ref).max() / np.abs(ref).max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_rank_gloo_bench_sharding_rows_and_checksum(scaling):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, scaling, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, "worker failed"
    res = [q.get(timeout=10) for _ in range(2)]
    for _, rel, err in res:
        assert rel < 1e-12 and err < 1e-12


def _inline_worker(rank, world, port, q):
    """InlineGridReducer (the schedule bench.py takes for short collectives) on two gloo ranks with CPU tensors: three
    steps over two buffers, rows-only reduction, every reduced grid against the one-process grid of the whole stream."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "ska-sdp-accelerate-gridding_amd", "python"))
    import torch
    import torch.distributed as dist
    import bench
    from gridhip.distributed import InlineGridReducer, mirrored_first_row, shard_bounds
    from oracle import gridref_c
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, W, Q, S, n = 160, 4, 4, 7, 4001
    dev = torch.device("cpu")
    gcf = bench.synth_kernels(W, Q, S, dev)
    bufs = [torch.zeros((N, N), dtype=torch.complex128) for _ in range(2)]
    red = InlineGridReducer(bufs, rows=(mirrored_first_row(N, S), N))
    errs = []
    for i in range(3):
        seed = 100 + i
        lo, hi = shard_bounds(n, world, rank)
        u, v, wb, vis = bench.synth_vis(hi - lo, N, W, S, seed, dev, lo=lo)
        g = red.begin(i)
        gridref_c.convgrid2(gcf.numpy(), g.numpy(), u.numpy(), v.numpy(), wb.numpy(), vis.numpy())
        red.end(i)
        U, V, WB, VIS = bench.synth_vis(n, N, W, S, seed, dev)
        ref = gridref_c.convgrid2(gcf.numpy(), np.zeros((N, N), dtype=np.complex128), U.numpy(), V.numpy(), WB.numpy(), VIS.numpy())
        errs.append(float(np.abs(g.numpy() - ref).max() / np.abs(ref).max()))
    red.finish()
    red.close()
    q.put((rank, max(errs)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_inline_reducer():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_inline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, "worker failed"
    for _ in range(2):
        assert q.get(timeout=10)[1] < 1e-12
