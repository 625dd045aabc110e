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


def mirrored_first_row(H, gh):
    """First grid row a mirrored stream (mirror_uvw, src/Gridding.hs:551-562: v >= 0) can touch: frac_coord puts
    v >= 0 at y >= H div 2 (floor (H/2 + v H + 0.5/Q) with v H >= 0) and the footprint starts gh div 2 rows below its
    centre (:170-171); one more row of margin.  Rows below it are exactly zero in every partial grid, so the
    all-reduce may skip them (half of its bytes)."""
    return max(0, H // 2 - gh // 2 - 1)


def allreduce_grid(grid, group=None, async_op=False, rows=None):
    """In-place fp64 sum of a complex128 grid over all ranks.

    torch tensor (cpu or cuda) or numpy array (wrapped without a copy).  rows = (y0, y1): only those rows (a
    contiguous slice of the row-major grid).  Returns the grid, or with async_op=True the torch.distributed work
    handle (its .wait() orders the current stream after the collective)."""
    import torch
    import torch.distributed as dist
    part = grid if rows is None else grid[rows[0]:rows[1]]
    if isinstance(part, np.ndarray):
        t = torch.from_numpy(part.view(np.float64))
    else:
        t = torch.view_as_real(part)
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else grid


class OverlappedGridReducer:
    """All-reduce of step i's grid on a side stream while step i+1 is gridded onto a second buffer
    (xGMI traffic hidden behind the tile kernel).  `grids` are two equally shaped cuda complex128
    tensors used alternately; call begin(i) before gridding onto grids[i % 2], end(i) after the
    gridding has been enqueued on the current stream, finish() before reading results."""

    def __init__(self, grids, group=None, rows=None):
        import torch
        self.torch = torch
        self.grids = grids
        self.group = group
        self.rows = rows  # (y0, y1): reduce these rows only (mirrored streams: mirrored_first_row)
        self.comm = torch.cuda.Stream(device=grids[0].device)
        self.work = [None, None]

    def begin(self, i, zero=True):
        """The buffer step i grids onto.  Its previous reduction is waited for and (zero=True) it is cleared:
        a step's all-reduce sums this step's partial grids, not the previous step's reduced content again."""
        w = self.work[i % 2]
        if w is not None:
            w.wait()  # this buffer's previous reduction must finish before it is written again
            self.work[i % 2] = None
        if zero:
            self.grids[i % 2].zero_()
        return self.grids[i % 2]

    def end(self, i):
        torch = self.torch
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ev)
            self.work[i % 2] = allreduce_grid(self.grids[i % 2], self.group, async_op=True, rows=self.rows)

    def finish(self):
        for k in (0, 1):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
        self.torch.cuda.current_stream().wait_stream(self.comm)

    def close(self):
        self.finish()


class InlineGridReducer:
    """The same begin(i) / end(i) / finish() protocol without overlap: step i's collective is enqueued right behind its
    gridding, on the gridding stream (libgridhip's communicator) or with the gridding stream waiting for it
    (torch.distributed), and the next step starts when it is done.  What a SHORT collective may want: a collective on a
    side stream shares the CUs and the memory system with the next step's gridding for as long as it runs
    (tools/pipeline_overlap_probe.py); bench.py --overlap auto tries this schedule against the side-stream ones and
    keeps the fastest."""

    def __init__(self, grids, group=None, rows=None, comm=None):
        self.grids, self.group, self.rows, self.c = grids, group, rows, comm

    def begin(self, i, zero=True):
        if zero:
            self.grids[i % len(self.grids)].zero_()
        return self.grids[i % len(self.grids)]

    def end(self, i):
        g = self.grids[i % len(self.grids)]
        if self.c is None:
            allreduce_grid(g, self.group, rows=self.rows)
        elif self.rows is None:
            self.c.allreduce_grid(g)
        else:
            self.c.allreduce_grid_rows(g, *self.rows)

    def finish(self):
        pass

    def close(self):
        pass


class OverlappedCommReducer:
    """OverlappedGridReducer over libgridhip's own communicator (Comm, rank form) instead of torch.distributed: the
    collective (option "collective": all-reduce, or reduce-scatter + all-gather) is enqueued by RCCL on a side stream
    handed to the communicator (gridhip_comm_set_stream), ordered after the step's gridding by an event and waited
    for, by event, before the buffer is written again.  Same protocol: begin(i) / end(i) / finish()."""

    def __init__(self, comm, grids, rows=None):
        import torch
        self.torch = torch
        self.c = comm
        self.grids = grids
        self.rows = rows
        self.side = torch.cuda.Stream(device=grids[0].device)
        comm.set_stream(self.side.cuda_stream)
        self.done = [None, None]

    def begin(self, i, zero=True):
        ev = self.done[i % 2]
        if ev is not None:
            self.torch.cuda.current_stream().wait_event(ev)  # this buffer's previous reduction
            self.done[i % 2] = None
        if zero:
            self.grids[i % 2].zero_()
        return self.grids[i % 2]

    def end(self, i):
        torch = self.torch
        ev = torch.cuda.Event()
        ev.record()  # after the step's gridding on the current stream
        self.side.wait_event(ev)
        g = self.grids[i % 2]
        if self.rows is None:
            self.c.allreduce_grid(g)
        else:
            self.c.allreduce_grid_rows(g, *self.rows)
        d = torch.cuda.Event()
        d.record(self.side)
        self.done[i % 2] = d

    def finish(self):
        for k in (0, 1):
            if self.done[k] is not None:
                self.torch.cuda.current_stream().wait_event(self.done[k])
                self.done[k] = None

    def close(self):
        self.finish()
        self.c.reset_stream()


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


class Comm:
    """libgridhip's own RCCL communicator (include/gridhip.h, gridhip_comm_*): the multi-GPU surface a
    non-Python host binds.  Comm.single_process(ndev) drives ndev devices from this process
    (ncclCommInitAll); Comm.from_torch(ctx) is the one-process-per-GPU form, the 128-byte id travelling
    over the already initialised torch.distributed group."""

    def __init__(self, handle, lib, ctxs=None):
        self._h, self._lib, self.ctxs = handle, lib, ctxs or []

    @staticmethod
    def _err(lib, rc, h=None):
        from ._lib import GridHipError
        return GridHipError(rc, (lib.gridhip_comm_last_error(h) or b"").decode())

    @classmethod
    def single_process(cls, ndev, dev_ids=None):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        h = C.c_void_p()
        ids = (C.c_int * ndev)(*dev_ids) if dev_ids is not None else None
        rc = lib.gridhip_comm_create(int(ndev), ids, C.byref(h))
        if rc != 0:
            raise cls._err(lib, rc)
        return cls(h, lib)

    @classmethod
    def from_torch(cls, ctx, group=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        lib = _lib.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = (C.c_char * 128)()
        if rank == 0:
            rc = lib.gridhip_comm_unique_id(buf)
            if rc != 0:
                raise cls._err(lib, rc)
        dev = torch.device("cuda", ctx.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(dev)
        dist.broadcast(t, src=0, group=group)
        ident = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        h = C.c_void_p()
        rc = lib.gridhip_comm_create_rank(ctx._h, world, rank, ident, C.byref(h))
        if rc != 0:
            raise cls._err(lib, rc)
        return cls(h, lib, [ctx])

    @property
    def ndev(self):
        return self._lib.gridhip_comm_ndev(self._h)

    @property
    def nranks(self):
        return self._lib.gridhip_comm_nranks(self._h)

    def allreduce_grid(self, grid):
        """In-place fp64 sum over the communicator of a cuda complex128 tensor (rank form), enqueued on the
        context's stream."""
        import ctypes as C
        rc = self._lib.gridhip_comm_allreduce_grid(self._h, grid.numel(), C.c_void_p(grid.data_ptr()))
        if rc != 0:
            raise self._err(self._lib, rc, self._h)
        return grid

    def allreduce_grid_rows(self, grid, y0, y1):
        """The same for rows [y0, y1) only (gridhip_comm_allreduce_grid_rows)."""
        import ctypes as C
        assert grid.dim() == 2 and grid.is_contiguous() and 0 <= y0 <= y1 <= grid.shape[0]
        rc = self._lib.gridhip_comm_allreduce_grid_rows(self._h, grid.shape[1], int(y0), int(y1), C.c_void_p(grid.data_ptr()))
        if rc != 0:
            raise self._err(self._lib, rc, self._h)
        return grid

    def set_option(self, key, value):
        rc = self._lib.gridhip_comm_set_option(self._h, key.encode(), int(value))
        if rc != 0:
            raise self._err(self._lib, rc, self._h)

    def get_option(self, key):
        import ctypes as C
        v = C.c_int64()
        rc = self._lib.gridhip_comm_get_option(self._h, key.encode(), C.byref(v))
        if rc != 0:
            raise self._err(self._lib, rc, self._h)
        return v.value

    def set_stream(self, stream_ptr, i=0):
        """Enqueue device i's collectives on this hipStream_t instead of its context's stream."""
        import ctypes as C
        rc = self._lib.gridhip_comm_set_stream(self._h, int(i), C.c_void_p(stream_ptr or 0))
        if rc != 0:
            raise self._err(self._lib, rc, self._h)

    def reset_stream(self, i=0):
        rc = self._lib.gridhip_comm_reset_stream(self._h, int(i))
        if rc != 0:
            raise self._err(self._lib, rc, self._h)

    def convgrid2(self, gcf, a, p, wbin, v):
        """convgrid2 over all devices of the communicator, numpy host arrays (gridhip_comm_convgrid2)."""
        import ctypes as C
        gcf = np.ascontiguousarray(gcf, dtype=np.complex128)
        u, vv = np.ascontiguousarray(p[0], dtype=np.float64), np.ascontiguousarray(p[1], dtype=np.float64)
        vis = np.ascontiguousarray(v, dtype=np.complex128)
        wb = None if wbin is None else np.ascontiguousarray(wbin, dtype=np.int64)
        assert a.dtype == np.complex128 and a.flags.c_contiguous
        W, Q, _, gh, gw = gcf.shape
        ptr = lambda x: None if x is None else C.c_void_p(x.ctypes.data)
        rc = self._lib.gridhip_comm_convgrid2(self._h, a.shape[0], a.shape[1], ptr(a), len(u), W, Q, gh, gw, ptr(gcf),
                                              ptr(u), ptr(vv), 1, ptr(wb), ptr(vis))
        if rc != 0:
            raise self._err(self._lib, rc, self._h)
        return a

    def close(self):
        if self._h:
            self._lib.gridhip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
