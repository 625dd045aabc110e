"""Visibility-sharded gridding across ranks (one process per GPU, torch.distributed).

Gridding is linear in the visibility set (G = sum_k footprint_k), so the path shards by
visibility with no data-path exchange until the end: each rank grids a contiguous range of the
stream onto a private N x N complex128 grid and ONE fp64 sum all-reduce of the 2*N*N doubles
combines the partial grids (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  The reference itself is single-process; this is new surface.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous, balanced [lo, hi) of rank's visibilities; the ranges tile [0, n) exactly."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_grid(grid, group=None, async_op=False):
    """In-place fp64 sum of a complex128 grid over all ranks.

    torch tensor (cpu or cuda) or numpy array (wrapped without a copy).  Returns the grid, or with
    async_op=True the torch.distributed work handle (its .wait() orders the current stream after
    the collective)."""
    import torch
    import torch.distributed as dist
    if isinstance(grid, np.ndarray):
        t = torch.from_numpy(grid.view(np.float64))
    else:
        t = torch.view_as_real(grid)
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else grid


class OverlappedGridReducer:
    """All-reduce of step i's grid on a side stream while step i+1 is gridded onto a second buffer
    (xGMI traffic hidden behind the tile kernel).  `grids` are two equally shaped cuda complex128
    tensors used alternately; call begin(i) before gridding onto grids[i % 2], end(i) after the
    gridding has been enqueued on the current stream, finish() before reading results."""

    def __init__(self, grids, group=None):
        import torch
        self.torch = torch
        self.grids = grids
        self.group = group
        self.comm = torch.cuda.Stream(device=grids[0].device)
        self.work = [None, None]

    def begin(self, i):
        w = self.work[i % 2]
        if w is not None:
            w.wait()  # this buffer's previous reduction must finish before it is written again
            self.work[i % 2] = None
        return self.grids[i % 2]

    def end(self, i):
        torch = self.torch
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ev)
            self.work[i % 2] = allreduce_grid(self.grids[i % 2], self.group, async_op=True)

    def finish(self):
        for k in (0, 1):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
        self.torch.cuda.current_stream().wait_stream(self.comm)


def sharded_convgrid2(gridder, gcf, a, p, wbin, v, rank, world, group=None, reduce=True):
    """convgrid2 (src/Gridding.hs:199-244) over a visibility-sharded stream.

    `gridder(gcf, a, (u, v, w), wbin, vis)` grids this rank's shard onto `a` (gridhip
    Context.convgrid2 on a GPU); `p`, `wbin`, `v` are the FULL arrays, each rank slices its own
    range.  Note `a` is accumulated into on every rank, so a non-zero starting grid must only be
    supplied on one rank."""
    u, vv = p[0], p[1]
    lo, hi = shard_bounds(len(u), world, rank)
    gridder(gcf, a, (u[lo:hi], vv[lo:hi], None), None if wbin is None else wbin[lo:hi], v[lo:hi])
    if reduce and world > 1:
        allreduce_grid(a, group)
    return a
