"""The N > 1 code on hardware.  A one-GPU box cannot show scaling, but it can run every line of the multi-GPU path
with a communicator of one rank: an `nccl` (= RCCL) process group drives OverlappedGridReducer (side stream, events,
two buffers), libgridhip's own communicator (gridhip_comm_*) all-reduces and grids, and bench.py runs its N > 1 branch.
Each case runs in a fresh child process so that the process group exists before anything touches the GPU."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PRELUDE = textwrap.dedent(f"""
    import os, sys
    sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "ska-sdp-accelerate-gridding_amd", "python"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import numpy as np, torch, torch.distributed as dist
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import gridhip
    from gridhip.distributed import OverlappedGridReducer, Comm, allreduce_grid
    from oracle import gridref_c
    rng = np.random.default_rng(7)
    N, W, Q, S, n = 192, 8, 4, 9, 60000
    gcf = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
    us = [rng.uniform(-0.5, 0.5, n) for _ in range(5)]
    vs = [rng.uniform(-0.5, 0.5, n) for _ in range(5)]
    wb = rng.integers(0, W, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    refs = [gridref_c.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), us[i], vs[i], wb, vis) for i in range(5)]
    T = lambda a, dt=None: torch.as_tensor(a, device=dev)
    ctx = gridhip.Context(0)
    relerr = lambda g, r: float(np.abs(g - r).max() / np.abs(r).max())
""")


def run_child(body):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, out.stdout[-2000:]
    return json.loads(lines[-1])


def test_overlapped_reducer_on_an_nccl_group_of_one():
    """five steps over two buffers: every reduced grid equals that step's plain convgrid2 (the buffer is cleared in
    begin(), the previous reduction is waited for, the side stream is ordered after the gridding)"""
    rec = run_child("""
        bufs = [torch.zeros((N, N), dtype=torch.complex128, device=dev) for _ in range(2)]
        red = OverlappedGridReducer(bufs)
        tg, tw, tv = T(gcf), T(wb), T(vis)
        errs, results = [], []
        for i in range(5):
            g = red.begin(i)
            ctx.convgrid2(tg, g, (T(us[i]), T(vs[i]), None), tw, tv)
            red.end(i)
            if i >= 1:  # the previous step's buffer: its reduction runs beside this step's gridding
                red.work[(i - 1) % 2].wait()
                errs.append(relerr(bufs[(i - 1) % 2].cpu().numpy(), refs[i - 1]))
        red.finish()
        errs.append(relerr(bufs[4 % 2].cpu().numpy(), refs[4]))
        import json; print(json.dumps({"errs": errs, "errors": ctx.get_option("errors")}))
        dist.destroy_process_group()
    """)
    assert len(rec["errs"]) == 5 and max(rec["errs"]) < 1e-10 and rec["errors"] == 0


def test_cabi_communicator_rank_form_and_single_process_form():
    rec = run_child("""
        import json
        # rank form: the id travels over the torch group; all-reduce of one rank leaves the grid unchanged
        comm = Comm.from_torch(ctx)
        g = torch.zeros((N, N), dtype=torch.complex128, device=dev)
        ctx.convgrid2(T(gcf), g, (T(us[0]), T(vs[0]), None), T(wb), T(vis))
        comm.allreduce_grid(g)
        ctx.synchronize()
        e_rank = relerr(g.cpu().numpy(), refs[0])
        sizes = (comm.ndev, comm.nranks)
        comm.close()
        # single-process form (what a Haskell host binds): shards, grids, reduces and accumulates INTO the host grid
        comm1 = Comm.single_process(1)
        start = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
        G = start.copy()
        comm1.convgrid2(gcf, G, (us[1], vs[1], None), wb, vis)
        e_single = relerr(G - start, refs[1])
        # n = 0 and bad arguments
        G0 = start.copy(); comm1.convgrid2(gcf, G0, (us[1][:0], vs[1][:0], None), wb[:0], vis[:0])
        same = bool(np.array_equal(G0, start))
        comm1.close()
        bad = None
        try:
            Comm.single_process(2, [0, 0])
        except gridhip.GridHipError as e:
            bad = e.code
        # a shard whose records are lost (test hook fault_inject on the communicator's own context): the synchronous
        # form reports the inconsistency instead of handing back an incomplete grid, and leaves the caller's grid alone
        import ctypes as C
        comm2 = Comm.single_process(1)
        h = comm2._lib.gridhip_comm_ctx(comm2._h, 0)
        assert comm2._lib.gridhip_set_option(C.c_void_p(h), b"fault_inject", 4000) == 0
        G2 = start.copy()
        lost = None
        try:
            comm2.convgrid2(gcf, G2, (us[1], vs[1], None), wb, vis)
        except gridhip.GridHipError as e:
            lost = (e.code, "consistency" in str(e), bool(np.array_equal(G2, start)))
        comm2.close()
        print(json.dumps({"e_rank": e_rank, "e_single": e_single, "sizes": sizes, "same": same, "bad": bad, "lost": lost}))
        dist.destroy_process_group()
    """)
    assert rec["e_rank"] < 1e-10 and rec["e_single"] < 1e-10
    assert rec["sizes"] == [1, 1] and rec["same"] and rec["bad"] == -1
    assert rec["lost"] == [-1, True, True]


def test_cabi_rows_reduce_scatter_and_side_stream_with_one_rank():
    """The pieces the 8-GPU run depends on, each with a communicator of one rank: the reduce-scatter + all-gather form
    of the collective, the rows-only form, both on a side stream handed to the communicator, driven by
    OverlappedCommReducer over five steps and two buffers while the tile kernel leaves CUs free (reserve_cus)."""
    rec = run_child("""
        import json
        from gridhip.distributed import OverlappedCommReducer, mirrored_first_row
        comm = Comm.from_torch(ctx)
        out = {"default": comm.get_option("collective")}
        try:
            comm.set_option("collective", 7)
            out["bad"] = 0
        except gridhip.GridHipError as e:
            out["bad"] = e.code
        try:
            comm.set_option("no_such_option", 1)
            out["bad2"] = 0
        except gridhip.GridHipError as e:
            out["bad2"] = e.code
        ctx.set_option("reserve_cus", 16)
        tg, tw, tv = T(gcf), T(wb), T(vis)
        errs = {}
        for name, coll, rows in (("ar_rows", 0, (40, N)), ("rs", 1, None), ("rs_rows", 1, (3, N - 5))):
            comm.set_option("collective", coll)
            bufs = [torch.zeros((N, N), dtype=torch.complex128, device=dev) for _ in range(2)]
            red = OverlappedCommReducer(comm, bufs, rows=rows)
            e = []
            for i in range(5):
                g = red.begin(i)
                ctx.convgrid2(tg, g, (T(us[i]), T(vs[i]), None), tw, tv)
                red.end(i)
                if i >= 1:
                    red.done[(i - 1) % 2].synchronize()
                    e.append(relerr(bufs[(i - 1) % 2].cpu().numpy(), refs[i - 1]))
            red.finish()
            torch.cuda.synchronize()
            e.append(relerr(bufs[0].cpu().numpy(), refs[4]))
            red.close()
            errs[name] = max(e)
        out["errs"] = errs
        out["collective"] = comm.get_option("collective")
        out["errors"] = ctx.get_option("errors")
        out["reserve"] = ctx.get_option("reserve_cus")
        # direct calls on the context's own stream again (reset by close())
        g = torch.zeros((N, N), dtype=torch.complex128, device=dev)
        ctx.convgrid2(tg, g, (T(us[0]), T(vs[0]), None), tw, tv)
        comm.allreduce_grid_rows(g, 0, N)
        comm.allreduce_grid_rows(g, 17, 17)   # empty range
        ctx.synchronize()
        out["direct"] = relerr(g.cpu().numpy(), refs[0])
        try:
            comm.allreduce_grid_rows(g, 5, N + 1)
            out["bad3"] = 0
        except AssertionError:
            out["bad3"] = 1
        comm.close()
        print(json.dumps(out))
        dist.destroy_process_group()
    """)
    assert rec["default"] == 0 and rec["bad"] == -1 and rec["bad2"] == -1 and rec["collective"] == 1 and rec["bad3"] == 1
    assert max(rec["errs"].values()) < 1e-10 and rec["direct"] < 1e-10 and rec["errors"] == 0 and rec["reserve"] == 16


@pytest.mark.parametrize("collective", ["torch", "cabi", "cabi-rs"])
def test_bench_multi_gpu_branch_with_one_rank(collective):
    """bench.py's N > 1 branch (process group, reducer / communicator, all-reduce timing, multi_gpu record)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(GRIDHIP_BENCH_FORCE_DIST="1", MASTER_PORT="29535")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--nvis",
                          "3000000", "--no-cpu", "--collective", collective, "--reserve-cus", "16", "--overlap",
                          "auto" if collective == "cabi" else "side"], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["multi_gpu"]["rccl_ranks"] == 1 and rec["multi_gpu"]["collective"] == collective
    assert rec["multi_gpu"]["allreduce_ms_alone"] > 0 and rec["errors"] == 0
    assert rec["check"]["rel_err"] <= 1e-10 and rec["multi_gpu"]["check_rel_err"] == rec["check"]["rel_err"]
    assert rec["multi_gpu"]["reduced_rows"] == [4096 // 2 - 15 // 2 - 1, 4096]
    assert rec["scaling"] == "weak" and rec["multi_gpu"]["scaling"] == "weak"
    sch = rec["multi_gpu"]["schedule"]
    side = sch["collective_runs"].startswith("side stream")
    assert rec["multi_gpu"]["reserve_cus"] == (16 if side else 0) and rec["multi_gpu"]["yield_cus"] == 0
    if collective == "cabi":  # auto with the reservation given: in line against the side stream with 16 CUs idle, measured
        assert sch["chosen_by"] == "measurement" and len(sch["tried_ms_per_step"]) == 2
        assert min(sch["tried_ms_per_step"].values()) > 0
    else:
        assert side and sch["chosen_by"] == "flags"
    assert 0 < rec["roofline"]["frac"] <= 1 and 0 < rec["roofline"]["lds_floor_frac"] <= 1
    assert rec["roofline"]["bound"] == "lds_atomic" and 1.0 < rec["roofline"]["clock_GHz"] < 2.6


def test_bench_default_line_is_bounded():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--nvis",
                          "4000000", "--cpu-sample", "200000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    r = rec["roofline"]
    assert 0 < r["frac"] <= 1 and 0 < r["lds_floor_frac"] <= 1 and r["kernel_ms"]["min"] <= r["kernel_ms"]["median"]
    assert rec["cpu_baseline"]["cores"] >= 1 and len(rec["cpu_baseline"]["modes"]) == 3
    assert rec["cpu_baseline"]["cpu_model"] and rec["value"] > 0
    # the line certifies itself: checksum of every grid the steps produced, and the GPU's grid of the CPU baseline's
    # sample against the oracle's, cell by cell
    c = rec["check"]
    assert c["rel_err"] <= 1e-10 and c["cells_nonzero"] > 1000 and c["grids_checked"] == 4 and c["seconds"] < 5.0
    assert rec["cpu_baseline"]["parity_rel_err"] <= 1e-10 and rec["errors"] == 0


def _bench(*extra, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", *extra],
                         capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out.returncode, (json.loads(lines[-1]) if lines else None), out.stderr


def test_bench_strong_scaling_on_one_gpu_is_the_weak_line():
    """With one GPU both modes draw the same range of the same global stream: same grids (cells_nonzero), same check."""
    rc_w, weak, err_w = _bench("--nvis", "3000000")
    rc_s, strong, err_s = _bench("--nvis", "3000000", "--scaling", "strong")
    assert rc_w == 0 and rc_s == 0, (err_w[-2000:], err_s[-2000:])
    assert weak["scaling"] == "weak" and strong["scaling"] == "strong"
    for k in ("vis_total", "vis_per_gpu", "seed"):
        assert weak["config"][k] == strong["config"][k]
    assert weak["check"]["cells_nonzero"] == strong["check"]["cells_nonzero"]
    assert weak["check"]["rel_err"] <= 1e-10 and strong["check"]["rel_err"] <= 1e-10


def test_bench_fails_loudly_when_the_result_is_wrong():
    """fault_inject hides record slots from the pre-pass: visibilities are lost, the library counts errors, the
    checksum is off - the process must exit non-zero, not print a number as if nothing had happened."""
    rc, rec, err = _bench("--nvis", "3000000", "--opt", "fault_inject=5000")
    assert rc != 0 and "RESULT CHECK FAILED" in err
    assert rec is not None and rec["errors"] > 0 and rec["check"]["rel_err"] > 1e-10


def test_two_ranks_share_one_gpu_over_gloo():
    """Two real rank processes (both on cuda:0 - NCCL refuses two ranks on one device, so the collective is gloo's, which
    stages cuda tensors through the host): each grids its shard of the stream through OverlappedGridReducer for four
    steps; every reduced grid must equal the oracle's grid of the whole stream.  This is the N > 1 control flow
    (sharding, buffer clearing, side-stream ordering, waiting on the previous reduction) with two ranks that
    really run concurrently."""
    body = textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "ska-sdp-accelerate-gridding_amd", "python"))
        import numpy as np, torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), 2
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        import gridhip
        from gridhip.distributed import OverlappedGridReducer, shard_bounds
        from oracle import gridref_c
        rng = np.random.default_rng(21)                      # the same stream on both ranks
        N, W, Q, S, n = 160, 4, 4, 7, 30001
        gcf = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
        steps = [(rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)) for _ in range(4)]
        wb = rng.integers(0, W, n)
        vis = rng.normal(size=n) + 1j * rng.normal(size=n)
        lo, hi = shard_bounds(n, world, rank)
        T = lambda a: torch.as_tensor(a, device=dev)
        ctx = gridhip.Context(0)
        bufs = [torch.zeros((N, N), dtype=torch.complex128, device=dev) for _ in range(2)]
        red = OverlappedGridReducer(bufs)
        errs = []
        def check(i):
            if red.work[i % 2] is not None:
                red.work[i % 2].wait()
            torch.cuda.synchronize()
            ref = gridref_c.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), steps[i][0], steps[i][1], wb, vis)
            errs.append(float(np.abs(bufs[i % 2].cpu().numpy() - ref).max() / np.abs(ref).max()))
        for i in range(4):
            g = red.begin(i)
            ctx.convgrid2(T(gcf), g, (T(steps[i][0][lo:hi]), T(steps[i][1][lo:hi]), None), T(wb[lo:hi]), T(vis[lo:hi]))
            red.end(i)
            if i >= 1:
                check(i - 1)
        red.finish()
        check(3)
        print(json.dumps({{"rank": rank, "errs": errs, "errors": ctx.get_option("errors")}}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
    """)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_RANK",)}
        env.update(RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", body], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    for so, _ in outs:
        rec = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
        assert len(rec["errs"]) == 4 and max(rec["errs"]) < 1e-10 and rec["errors"] == 0


@pytest.mark.parametrize("scaling,overlap", [("weak", "side"), ("strong", "inline"), ("strong", "auto")])
def test_bench_with_two_real_ranks_sharing_the_gpu(scaling, overlap):
    """bench.py itself with two rank processes (both on cuda:0, gloo's host-staged collectives in place of RCCL, which
    refuses two ranks on one device): the global stream sharded weak and strong, the untimed scheduling pass, the
    side-stream and in-line reducers with the rows-only reduction, and the line's own verdict - the REDUCED grid of two
    real shards against the checksum all-reduced over the ranks (a rank gridding the wrong range, a buffer reduced
    twice or cleared late, rows skipped that were not zero: all show there)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GRIDHIP_BENCH_SHARE_GPU="1", GRIDHIP_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--nvis", "3000000", "--scaling", scaling, "--overlap", overlap], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == scaling and rec["errors"] == 0
    assert rec["config"]["vis_total"] == (6000000 if scaling == "weak" else 3000000)
    assert rec["config"]["vis_per_gpu"] == (3000000 if scaling == "weak" else 1500000)
    assert rec["check"]["rel_err"] <= 1e-10 and rec["multi_gpu"]["check_rel_err"] <= 1e-10
    assert rec["check"]["cells_nonzero"] > 100000
    m = rec["multi_gpu"]
    assert m["rccl_ranks"] == 2 and m["reduced_rows"] == [2040, 4096] and m["allreduce_bytes"] == (4096 - 2040) * 4096 * 16
    assert m["schedule"]["collective_runs"].startswith("side" if overlap == "side" else "in line") or overlap == "auto"
    tried = m["schedule"]["tried_ms_per_step"]
    assert m["schedule"]["chosen_by"] == ("flags" if overlap == "inline" else "measurement")
    assert tried is None if overlap == "inline" else len(tried) == (3 if overlap == "side" else 4)
    assert (m["reserve_cus"], m["yield_cus"]) in ((0, 0), (0, 64), (32, 0))
    assert m["schedule"]["collective_runs"].startswith("side") or (m["reserve_cus"], m["yield_cus"]) == (0, 0)
