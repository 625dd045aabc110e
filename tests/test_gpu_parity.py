"""GPU parity: libgridhip (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): grids within 1e-10 relative of the CPU reference,
measured as max|G_gpu - G_ref| / max|G_ref|.  fp64 atomics make the summation order differ
from the oracle's, so results are a tolerance, never bit-exact; observed error is ~1e-15.
Integer work (cell / sub-cell indices) must match exactly — it shows up as O(1) grid errors
when it does not.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL = 1e-10


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def case(seed, N, M, W, Q, gh, gw, n, spread=0.56, dist="uniform"):
    rng = np.random.default_rng(seed)
    gcf = rng.normal(size=(W, Q, Q, gh, gw)) + 1j * rng.normal(size=(W, Q, Q, gh, gw))
    if dist == "uniform":
        u = rng.uniform(-spread, spread, n)
        v = rng.uniform(-spread, spread, n)
    else:  # centrally concentrated: heavy same-tile contention
        u = np.clip(rng.normal(0, 0.03, n), -0.6, 0.6)
        v = np.clip(rng.normal(0, 0.03, n), -0.6, 0.6)
    wb = rng.integers(0, W, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    return gcf, u, v, wb, vis


SHAPES = [
    # N(rows) M(cols) W  Q  gh  gw   n
    (64, 64, 3, 4, 7, 7, 3000),
    (256, 256, 8, 8, 15, 15, 20000),
    (128, 96, 2, 2, 5, 9, 5000),        # non-square grid, non-square kernel
    (96, 160, 4, 3, 9, 5, 5000),
    (200, 200, 2, 2, 31, 31, 2000),     # the support the reference's test scripts use (S=31)
    (64, 64, 1, 1, 1, 1, 1000),
    (50, 50, 5, 2, 3, 3, 4000),
    (33, 47, 2, 4, 2, 4, 2000),         # even supports, odd grid sizes
    (300, 300, 16, 8, 15, 15, 50000),
    (160, 160, 4, 4, 9, 9, 20000),      # the other compile-time supports of the pipelined kernel
    (160, 160, 4, 4, 11, 11, 20000),
    (160, 160, 4, 2, 13, 13, 20000),
    (160, 160, 4, 4, 5, 5, 20000),
    (160, 160, 4, 2, 6, 6, 20000),      # even supports have instantiations too
    (160, 160, 4, 2, 12, 12, 20000),
    (160, 160, 2, 2, 16, 16, 20000),
]


@pytest.mark.parametrize("N,M,W,Q,gh,gw,n", SHAPES)
def test_convgrid2_matches_oracle(ctx, oracle, N, M, W, Q, gh, gw, n):
    gcf, u, v, wb, vis = case(N * 7 + gw, N, M, W, Q, gh, gw, n)
    ref = oracle.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), u, v, wb, vis)
    got = ctx.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), (u, v, None), wb, vis)
    assert rel(got, ref) < TOL
    assert ctx.last_dropped() == 0


@pytest.mark.parametrize("tile,block,wgroups,chunk,sort", [(8, 64, 1, 64, 0), (16, 256, 2, 100, 0), (32, 512, 4, 512, 0),
                                                           (64, 1024, 8, 4096, 2), (64, 256, 1, 0, 2), (32, 1024, 8, 64, 0),
                                                           (64, 1024, 8, 0, 1), (64, 1024, 1, 0, 1), (32, 512, 4, 700, 1),
                                                           (64, 256, 2, 0, 1), (16, 128, 8, 0, 1),
                                                           (64, 64, 2, 0, 1),     # one wave: sorter and walker in turn
                                                           (64, 128, 8, 300, 1),  # one walker + the sorter
                                                           (32, 1024, 1, 16384, 1)])
def test_tuning_knobs_do_not_change_results(ctx, oracle, tile, block, wgroups, chunk, sort):
    N, W, Q, S, n = 200, 8, 4, 15, 30000
    gcf, u, v, wb, vis = case(99, N, N, W, Q, S, S, n)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis)
    try:
        for k, val in (("tile", tile), ("block", block), ("wgroups", wgroups), ("chunk", chunk), ("sort", sort)):
            ctx.set_option(k, val)
        got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
    finally:
        for k in ("tile", "block", "wgroups", "chunk", "sort"):
            ctx.set_option(k, 0)
    assert rel(got, ref) < TOL


@pytest.mark.parametrize("dist", ["uniform", "core"])
def test_sorted_variant_matches_oracle(ctx, oracle, dist):
    """The tap-reusing kernel (records sorted by kernel slice in LDS, runs share registers)."""
    for (N, W, Q, S, n) in [(512, 32, 8, 15, 150000), (300, 16, 4, 7, 80000), (256, 8, 4, 9, 60000),
                            (256, 8, 4, 11, 60000), (256, 8, 2, 13, 60000), (200, 4, 4, 5, 50000),
                            (256, 8, 2, 8, 60000), (256, 8, 2, 14, 60000), (256, 4, 2, 16, 60000)]:
        gcf, u, v, wb, vis = case(7 + N, N, N, W, Q, S, S, n, dist=dist)
        ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
        ctx.set_option("sort", 1)
        try:
            got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
        finally:
            ctx.set_option("sort", 0)
        assert rel(got, ref) < TOL


@pytest.mark.parametrize("shape", [(256, 256, 8, 8, 15, 15, 20000, {}), (128, 96, 2, 2, 5, 9, 5000, {}),
                                   (300, 300, 16, 8, 15, 15, 150000, {"sort": 1}),
                                   (1024, 1024, 8, 2, 7, 7, 200000, {"tile": 16, "wgroups": 8}),  # 34 848 bins: 35 per coarse bin
                                   (2048, 2048, 8, 2, 7, 7, 300000, {"tile": 32, "wgroups": 8}),
                                   (2048, 2048, 8, 2, 7, 7, 300000, {"tile": 16, "wgroups": 8}),  # 4 counting windows, 136 coarse-bin width
                                   (2048, 2048, 8, 2, 7, 7, 300000, {"tile": 8, "wgroups": 4}),  # 264 196 bins > 2^18: 16-byte intermediate records
                                   (64, 64, 1, 1, 1, 1, 1000, {}), (200, 200, 2, 2, 31, 31, 2000, {})])
@pytest.mark.parametrize("dist,mode", [("uniform", 2), ("core", 2), ("uniform", 4), ("core", 5), ("uniform", 6)])
def test_two_level_prepass(ctx, oracle, shape, dist, mode):
    """The scatter of the binning pre-pass in two levels (LDS-sorted runs into coarse bins, then to the bins):
    automatic from 2^22 visibilities, forced here on small streams.  Coordinates spill over the grid edges
    (dropped visibilities leave holes in neither level) and some wbins are out of range.
    mode 2: the scatter reads the counting sweep's 8-byte pre-records and moves 8-byte records (the bin's index
    inside its coarse bin rides in the word's spare bits, the coarse bin follows from the record's position);
    4: it recomputes from the stream; 5, 6: 16- and 12-byte intermediate records (what streams whose fields leave no
    spare bits use)."""
    N, M, W, Q, gh, gw, n, opts = shape
    gcf, u, v, wb, vis = case(1234 + N, N, M, W, Q, gh, gw, n, spread=0.6, dist=dist)
    wb = wb.copy()
    wb[::97] = W + 3      # dropped and counted
    wb[5::101] = -1
    keep = (wb >= 0) & (wb < W)
    ref = oracle.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), u[keep], v[keep], wb[keep], vis[keep], mt_mode=2)
    try:
        ctx.set_option("prepass", mode)
        for k, val in opts.items():
            ctx.set_option(k, val)
        got = ctx.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), (u, v, None), wb, vis)
        dropped = ctx.last_dropped()
        assert ctx.get_option("errors") == 0
        d = ctx.degrid2(gcf, ref, (u, v, None), wb)
        ctx.set_option("prepass", 1)
        ctx.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), (u, v, None), wb, vis)
        dropped_one_level = ctx.last_dropped()
    finally:
        for k in ("prepass", "tile", "wgroups", "sort"):
            ctx.set_option(k, 0)
    assert rel(got, ref) < TOL
    # (a bad wbin is only counted when the coordinates are inside the grid)
    assert 0 < dropped <= int((~keep).sum()) and dropped == dropped_one_level
    dref = oracle.degrid2(gcf, ref, u[keep], v[keep], wb[keep])
    assert rel(d[keep], dref) < TOL
    assert np.all(d[~keep] == 0)


@pytest.mark.parametrize("shift", [0, 1])
def test_counting_sweep_with_and_without_16_byte_alignment(ctx, oracle, shift):
    """The counting sweep of the two-level pre-pass reads u, v, wbin two visibilities at a time (16-byte accesses,
    grid-stride) when the arrays are 16-byte aligned, and one at a time otherwise: device arrays that start 8 bytes
    into an allocation take the second path.  An odd stream length exercises the first path's tail."""
    import torch
    N, W, Q, S, n = 256, 8, 4, 9, 150001
    gcf, u, v, wb, vis = case(909, N, N, W, Q, S, S, n, spread=0.55)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    dev = torch.device("cuda:0")
    def put(a):  # a device copy whose first element sits `shift` elements into its allocation
        buf = torch.empty(len(a) + shift, dtype=torch.from_numpy(a[:1]).dtype, device=dev)
        buf[shift:] = torch.from_numpy(a).to(dev)
        return buf[shift:]
    tu, tv, twb = put(u), put(v), put(wb)
    assert tu.data_ptr() % 16 == 8 * shift
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    try:
        ctx.set_option("prepass", 2)
        ctx.convgrid2(torch.from_numpy(gcf).to(dev), G, (tu, tv, None), twb, torch.from_numpy(vis).to(dev))
        torch.cuda.synchronize()
        errors = ctx.get_option("errors")
    finally:
        ctx.set_option("prepass", 0)
    assert errors == 0 and rel(G.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("prepass", [1, 2])
def test_record_fields_that_fill_the_word_exactly(ctx, oracle, prepass):
    """Every part of a call that had to be cut has records whose fields take all 64 bits.  The test hook
    rec_bits = 164 widens the kernel-slice field to that point on a small case: no record may be reported as
    inconsistent (the host-pointer forms would return GRIDHIP_EINVAL), and the 12-byte intermediate records that
    streams without spare bits use are exercised on the way."""
    N, W, Q, S, n = 128, 8, 4, 7, 40000
    gcf, u, v, wb, vis = case(616, N, N, W, Q, S, S, n, spread=0.55)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    rng = np.random.default_rng(5)
    G = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    dref = oracle.degrid2(gcf, G, u, v, wb)
    try:
        ctx.set_option("prepass", prepass)
        ctx.set_option("rec_bits", 164)
        got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
        d = ctx.degrid2(gcf, G, (u, v, None), wb)
        errors = ctx.get_option("errors")
    finally:
        ctx.set_option("prepass", 0)
        ctx.set_option("rec_bits", 0)
    assert errors == 0 and rel(got, ref) < TOL and rel(d, dref) < TOL


@pytest.mark.parametrize("prepass", [1, 2])
def test_calls_whose_record_fields_do_not_fit_are_cut_into_parts(ctx, oracle, prepass):
    """A record is one 64-bit word (14 bits of footprint origin, the kernel slice, the visibility's index); a call
    with slices x visibilities above 2^50 is gridded / degridded in parts.  The test hook "rec_bits" pretends the
    word has 30 bits, so that this case (7 bits of slice, 16 of index) is cut into parts of 512 visibilities."""
    N, W, Q, S, n = 128, 8, 4, 7, 40000
    gcf, u, v, wb, vis = case(515, N, N, W, Q, S, S, n, spread=0.55)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    rng = np.random.default_rng(4)
    G = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    dref = oracle.degrid2(gcf, G, u, v, wb)
    try:
        ctx.set_option("prepass", prepass)
        ctx.set_option("rec_bits", 30)
        got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
        d = ctx.degrid2(gcf, G, (u, v, None), wb)
        errors = ctx.get_option("errors")
    finally:
        ctx.set_option("prepass", 0)
        ctx.set_option("rec_bits", 0)
    assert errors == 0 and rel(got, ref) < TOL and rel(d, dref) < TOL


@pytest.mark.parametrize("opts", [{"tile": 16, "wgroups": 4}, {"tile": 16, "wgroups": 8}, {"tile": 8, "wgroups": 8}])
def test_many_bins_windowed_prepass(ctx, oracle, opts):
    """More bins than one LDS histogram holds: the pre-pass covers them in windows (2 and 4 windows here),
    and past 8 windows counts in global memory (the 8x8-tile case)."""
    N, W, Q, S, n = 2048, 8, 2, 7, 300000
    gcf, u, v, wb, vis = case(31, N, N, W, Q, S, S, n, spread=0.5)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    try:
        for k, val in opts.items():
            ctx.set_option(k, val)
        got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
    finally:
        for k in opts:
            ctx.set_option(k, 0)
    assert rel(got, ref) < TOL


def test_direct_variant_matches_oracle(ctx, oracle):
    N, W, Q, S, n = 128, 4, 4, 7, 20000
    gcf, u, v, wb, vis = case(5, N, N, W, Q, S, S, n)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis)
    ctx.set_option("variant", 1)
    try:
        got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
    finally:
        ctx.set_option("variant", 0)
    assert rel(got, ref) < TOL


def test_concentrated_distribution(ctx, oracle):
    """Distribution B of SURVEY.md §8d: everything lands in a handful of tiles."""
    N, W, Q, S, n = 512, 8, 8, 15, 60000
    gcf, u, v, wb, vis = case(21, N, N, W, Q, S, S, n, dist="core")
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis)
    got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
    assert rel(got, ref) < TOL


def test_accumulates_into_existing_grid(ctx, oracle):
    """permute (+) a: the destination is added to, not overwritten (src/Gridding.hs:244)."""
    N, W, Q, S, n = 96, 2, 2, 7, 4000
    gcf, u, v, wb, vis = case(3, N, N, W, Q, S, S, n)
    rng = np.random.default_rng(4)
    G0 = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    ref = oracle.convgrid2(gcf, G0.copy(), u, v, wb, vis)
    got = ctx.convgrid2(gcf, G0.copy(), (u, v, None), wb, vis)
    assert rel(got, ref) < TOL
    # twice more onto the same grid
    ref = oracle.convgrid2(gcf, ref, u, v, wb, vis)
    got = ctx.convgrid2(gcf, got, (u, v, None), wb, vis)
    assert rel(got, ref) < TOL


def test_uvw_matrix_layout(ctx, oracle):
    """(n,3) row-major /vis/uvw matrix sliced by column (src/ImageDataset.hs:51-53, 94-97)."""
    N, W, Q, S, n = 96, 2, 2, 7, 4000
    gcf, u, v, wb, vis = case(8, N, N, W, Q, S, S, n)
    uvw = np.stack([u, v, np.zeros(n)], axis=1)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis)
    got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), uvw, wb, vis)
    assert rel(got, ref) < TOL


def test_edges_empty_and_out_of_range(ctx, oracle):
    N, W, Q, S = 64, 2, 2, 7
    gcf, u, v, wb, vis = case(12, N, N, W, Q, S, S, 16)
    z = lambda: np.zeros((N, N), dtype=np.complex128)
    # n = 0
    assert not ctx.convgrid2(gcf, z(), (u[:0], v[:0], None), wb[:0], vis[:0]).any()
    # everything far outside: dropped, never wrapped (fixoutofbounds, src/Gridding.hs:883-891)
    far = np.array([0.9, -0.9, 3.0, -1e6, 1e300, -1e300])
    assert not ctx.convgrid2(gcf, z(), (far, far[::-1].copy(), None), np.zeros(6, np.int64), np.ones(6, complex)).any()
    # NaN coordinates are skipped
    nanu = np.array([np.nan, 0.1, np.inf])
    nanv = np.array([0.1, np.nan, 0.0])
    assert not ctx.convgrid2(gcf, z(), (nanu, nanv, None), np.zeros(3, np.int64), np.ones(3, complex)).any()
    # footprints straddling each edge and corner: only the in-range part is filled
    eu = np.array([-0.5, 0.4999, 0.0, 0.0, -0.5, 0.4999, -0.53, 0.54])
    ev = np.array([0.0, 0.0, -0.5, 0.4999, -0.5, 0.4999, 0.2, -0.3])
    ewb = np.array([0, 1, 0, 1, 0, 1, 0, 1])
    evis = np.arange(1, 9) * (1 + 0.5j)
    ref = oracle.convgrid2(gcf, z(), eu, ev, ewb, evis)
    got = ctx.convgrid2(gcf, z(), (eu, ev, None), ewb, evis)
    assert ref.any() and rel(got, ref) < TOL
    # wbin outside [0, W): skipped and counted (the reference would read out of range)
    bad = np.array([0, W, -1, 1])
    got = ctx.convgrid2(gcf, z(), (u[:4], v[:4], None), bad, vis[:4])
    ref = oracle.convgrid2(gcf, z(), u[[0, 3]], v[[0, 3]], bad[[0, 3]], vis[[0, 3]])
    assert rel(got, ref) < TOL and ctx.last_dropped() == 2


def test_index_parity_near_boundaries(ctx, oracle):
    """Coordinates a hair away from cell / sub-cell boundaries must pick the oracle's cell."""
    N, Q, S = 64, 8, 3
    rng = np.random.default_rng(33)
    gcf = rng.normal(size=(1, Q, Q, S, S)) + 1j * rng.normal(size=(1, Q, Q, S, S))
    k = rng.integers(-30 * Q, 30 * Q, 4000)
    # sub-cell boundaries sit at (k + 0.5)/Q cells; nudge by +-1e-9 cells
    base = (k + 0.5) / Q
    pu = (base + rng.choice([-1e-9, 1e-9], 4000)) / N
    pv = rng.uniform(-0.4, 0.4, 4000)
    vis = rng.normal(size=4000) + 1j * rng.normal(size=4000)
    wb = np.zeros(4000, np.int64)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), pu, pv, wb, vis)
    got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (pu, pv, None), wb, vis)
    assert rel(got, ref) < TOL


def test_convgrid_and_grid(ctx, oracle):
    N, Q, S, n = 128, 4, 7, 10000
    gcf, u, v, wb, vis = case(17, N, N, 1, Q, S, S, n)
    ref = oracle.convgrid(gcf[0], np.zeros((N, N), dtype=np.complex128), u, v, vis)
    got = ctx.convgrid(gcf[0], np.zeros((N, N), dtype=np.complex128), (u, v, None), vis)
    assert rel(got, ref) < TOL
    ref = oracle.grid(np.zeros((N, N), dtype=np.complex128), u, v, vis)
    got = ctx.grid(np.zeros((N, N), dtype=np.complex128), (u, v, None), vis)
    assert rel(got, ref) < TOL


def test_golden_fixtures(ctx, golden):
    g = golden("brokennumbers")  # the reference's own recorded output
    G = np.zeros((5, 5), dtype=np.complex128)
    for _ in range(int(g["passes"])):
        ctx.grid(G, ((g["x"] - 2) / 5.0, (g["y"] - 2) / 5.0, None), g["val"])
    assert np.array_equal(G, g["expected"])
    g = golden("brokennumbers_real")  # ... and its real-valued twin (old/BrokenNumbers.hs:101-106)
    G = np.zeros((5, 5), dtype=np.complex128)
    for _ in range(int(g["passes"])):
        ctx.grid(G, ((g["x"] - 2) / 5.0, (g["y"] - 2) / 5.0, None), g["val"].astype(np.complex128))
    assert np.array_equal(G.real, g["expected"]) and not G.imag.any()
    g = golden("fixbounds")  # test/GridTesting.hs:365-387: the four offsets of a 2x2 footprint of ones
    G = ctx.convgrid(g["gcf"], np.zeros((5, 5), dtype=np.complex128), (g["pu"], g["pv"], None), g["vis"])
    assert np.array_equal(G, g["expected"])
    g = golden("fixbounds2")
    G = ctx.convgrid(g["gcf"], np.zeros((5, 5), dtype=np.complex128), (g["pu"], g["pv"], None), g["vis"])
    assert np.array_equal(G, g["expected"])
    g = golden("convgrid2_small")
    N = g["expected"].shape[0]
    G = ctx.convgrid2(g["gcf"], np.zeros((N, N), dtype=np.complex128), (g["u"], g["v"], None), g["wbin"], g["vis"])
    assert rel(G, g["expected"]) < TOL
    d = ctx.degrid2(g["gcf"], g["expected"].copy(), (g["u"], g["v"], None), g["wbin"])
    assert rel(d, g["degrid"]) < TOL


@pytest.mark.parametrize("N,M,W,Q,gh,gw,n", SHAPES[:5])
def test_degrid2_matches_oracle(ctx, oracle, N, M, W, Q, gh, gw, n):
    gcf, u, v, wb, vis = case(N + gw, N, M, W, Q, gh, gw, n)
    rng = np.random.default_rng(1)
    G = rng.normal(size=(N, M)) + 1j * rng.normal(size=(N, M))
    ref = oracle.degrid2(gcf, G, u, v, wb)
    got = ctx.degrid2(gcf, G, (u, v, None), wb)
    assert rel(got, ref) < TOL


@pytest.mark.parametrize("prepass", [0, 2])
def test_degrid2_writes_zero_for_dropped_visibilities(ctx, oracle, prepass):
    """degrid2 does not clear its output array: the counting sweep writes the zero prediction of every visibility
    it drops (no tap inside the grid, NaN coordinates, wbin outside [0, W)) and the tile kernel writes the rest.
    The output starts as NaN here, so an element nobody wrote would show.  Also through plans: with drops (the
    plan clears), without (it does not)."""
    import torch
    N, W, Q, S, n = 256, 8, 4, 9, 50000
    gcf, u, v, wb, vis = case(404, N, N, W, Q, S, S, n, spread=0.62)
    u[5], v[7] = np.nan, np.inf
    wb[11], wb[13] = -1, W
    rng = np.random.default_rng(8)
    G = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    keep = (wb >= 0) & (wb < W) & np.isfinite(u) & np.isfinite(v)
    ref = np.zeros(n, dtype=np.complex128)
    ref[keep] = oracle.degrid2(gcf, G, u[keep], v[keep], wb[keep])
    assert (ref == 0).sum() > 100      # the case really has dropped visibilities
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    nan = lambda: torch.full((n,), float("nan"), dtype=torch.complex128, device=dev)
    try:
        ctx.set_option("prepass", prepass)
        out = nan()
        ctx.degrid2(t(gcf), t(G), (t(u), t(v), None), t(wb), out)
        plan = ctx.plan((N, N), gcf.shape, (t(u), t(v), None), t(wb))
        pout = nan()
        plan.degrid(t(gcf), t(G), pout)
        plan.close()
        inside = np.abs(u) < 0.4
        inside &= (np.abs(v) < 0.4) & keep
        plan = ctx.plan((N, N), gcf.shape, (t(u[inside]), t(v[inside]), None), t(wb[inside]))
        qout = torch.full((int(inside.sum()),), float("nan"), dtype=torch.complex128, device=dev)
        plan.degrid(t(gcf), t(G), qout)
        plan.close()
    finally:
        ctx.set_option("prepass", 0)
    for got in (out.cpu().numpy(), pout.cpu().numpy()):
        assert np.isfinite(got).all() and rel(got, ref) < TOL
        assert not got[ref == 0].any()
    q = qout.cpu().numpy()
    assert np.isfinite(q).all() and rel(q, ref[inside]) < TOL


@pytest.mark.parametrize("N,M,W,Q,gh,gw,n,opts", [
    (320, 320, 8, 4, 17, 17, 60000, {}),                # 2 x 2 parts of 9
    (320, 320, 8, 2, 21, 21, 60000, {}),                # 2 x 2 parts of 11
    (384, 384, 4, 2, 31, 31, 40000, {}),                # 2 x 2 parts of 16: the support of the reference's scripts
    (384, 320, 4, 2, 31, 31, 40000, {"tile": 32, "wgroups": 2}),
    (256, 256, 8, 4, 9, 5, 60000, {}),                  # one 9 x 9 part, zero-padded
    (256, 192, 4, 2, 5, 20, 50000, {}),                 # 1 x 2 parts of 10
    (256, 256, 2, 2, 40, 33, 20000, {}),                # 3 x 3 parts of 14
    (200, 200, 4, 4, 3, 3, 50000, {}),                  # below the smallest instantiation: padded to 5 x 5
    (128, 128, 1, 1, 1, 1, 30000, {}),
])
@pytest.mark.parametrize("dist", ["uniform", "core"])
@pytest.mark.parametrize("subfoot", [0, 1])
def test_subfootprints_large_and_nonsquare_supports(ctx, oracle, N, M, W, Q, gh, gw, n, opts, dist, subfoot):
    """Supports above 16 and non-square kernels through the tap-reusing kernel.  Square supports 17..32 (subfoot = 0):
    one record per visibility, the kernel takes a slice's taps in parts of at most four steps.  Everything else, and
    all of them under option "subfoot" = 1 (round 2's cut, parts of side <= 16): the kernel table is cut into
    zero-padded square parts and every visibility becomes one record per part (own footprint origin, tile and slice).
    Grid, degrid (a visibility's parts are summed), a plan, and coordinates spilling over the grid edges."""
    import torch
    gcf, u, v, wb, vis = case(N + gh * 3 + gw, N, M, W, Q, gh, gw, n, spread=0.58, dist=dist)
    ref = oracle.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    rng = np.random.default_rng(9)
    G = rng.normal(size=(N, M)) + 1j * rng.normal(size=(N, M))
    dref = oracle.degrid2(gcf, G, u, v, wb)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    try:
        ctx.set_option("sort", 1)
        ctx.set_option("subfoot", subfoot)
        for k, val in opts.items():
            ctx.set_option(k, val)
        got = ctx.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), (u, v, None), wb, vis)
        e1 = ctx.get_option("errors")
        d = ctx.degrid2(gcf, G, (u, v, None), wb)
        e2 = ctx.get_option("errors")
        plan = ctx.plan((N, M), gcf.shape, (t(u), t(v), None), t(wb))
        pg = plan.grid(t(gcf), torch.zeros((N, M), dtype=torch.complex128, device=dev), t(vis)).cpu().numpy()
        pd = plan.degrid(t(gcf), t(G)).cpu().numpy()
        plan.close()
        ctx.set_option("prepass", 2)   # the two-level scatter with pre-records, one record per part
        got2 = ctx.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), (u, v, None), wb, vis)
    finally:
        for k in ("sort", "tile", "block", "wgroups", "prepass", "subfoot"):
            ctx.set_option(k, 0)
    assert e1 == 0 and e2 == 0
    assert rel(got, ref) < TOL and rel(got2, ref) < TOL and rel(pg, ref) < TOL
    assert rel(d, dref) < TOL and rel(pd, dref) < TOL


@pytest.mark.parametrize("S", list(range(17, 33)))
def test_every_support_17_to_32_in_parts_of_the_tap_list(ctx, oracle, S):
    """Each square support the tap-reusing kernel is instantiated for above 16 x 16 (a slice's steps in 2..4 parts,
    one record per visibility): grid, degrid and a plan against the oracle, dense enough that runs of equal slice hold
    several visibilities (the pair path of a 32-tap last step, the per-block extra taps of 17 x 17 and 31 x 31), with
    footprints spilling over every grid edge, on a non-square grid."""
    import torch
    N, M, W, Q, n = 288, 352, 3, 2, 70000
    gcf, u, v, wb, vis = case(1000 + S, N, M, W, Q, S, S, n, spread=0.6)
    ref = oracle.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    rng = np.random.default_rng(S)
    G = rng.normal(size=(N, M)) + 1j * rng.normal(size=(N, M))
    dref = oracle.degrid2(gcf, G, u, v, wb)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    try:
        ctx.set_option("sort", 1)
        got = ctx.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), (u, v, None), wb, vis)
        d = ctx.degrid2(gcf, G, (u, v, None), wb)
        errors = ctx.get_option("errors")
        plan = ctx.plan((N, M), gcf.shape, (t(u), t(v), None), t(wb))
        pd = plan.degrid(t(gcf), t(G)).cpu().numpy()
        pd2 = plan.degrid(t(gcf), t(G), out=torch.full((n,), 7.0 + 1j, dtype=torch.complex128, device=dev)).cpu().numpy()
        plan.close()
        ctx.set_option("wgroups", 1)
        ctx.set_option("tile", 24)
        got_small = ctx.convgrid2(gcf, np.zeros((N, M), dtype=np.complex128), (u, v, None), wb, vis)
    finally:
        for k in ("sort", "wgroups", "tile"):
            ctx.set_option(k, 0)
    assert errors == 0
    assert rel(got, ref) < TOL and rel(got_small, ref) < TOL
    assert rel(d, dref) < TOL and rel(pd, dref) < TOL and rel(pd2, dref) < TOL


@pytest.mark.parametrize("N,W,Q,S,n,opts", [(512, 32, 8, 15, 150000, {}), (300, 16, 4, 7, 80000, {}),
                                             (256, 8, 4, 9, 60000, {}), (256, 8, 2, 13, 60000, {}),
                                             (512, 32, 8, 15, 150000, {"tile": 64, "wgroups": 8}),
                                             (256, 8, 4, 15, 60000, {"tile": 32, "block": 256, "wgroups": 2})])
def test_degrid2_sorted_variant(ctx, oracle, N, W, Q, S, n, opts):
    """degrid2 through the tap-reusing kernel (tile loaded into LDS once, runs share taps)."""
    gcf, u, v, wb, vis = case(N + S, N, N, W, Q, S, S, n)
    rng = np.random.default_rng(5)
    G = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    ref = oracle.degrid2(gcf, G, u, v, wb)
    try:
        ctx.set_option("sort", 1)
        for k, val in opts.items():
            ctx.set_option(k, val)
        got = ctx.degrid2(gcf, G, (u, v, None), wb)
    finally:
        for k in ("sort", "tile", "block", "wgroups"):
            ctx.set_option(k, 0)
    assert ctx.get_option("errors") == 0
    assert rel(got, ref) < TOL


def test_adjointness_on_device(ctx):
    """<g, grid(vis)> == <degrid_{conj K}(g), vis> — ties the two kernels together."""
    N, W, Q, S, n = 160, 4, 4, 9, 20000
    gcf, u, v, wb, vis = case(77, N, N, W, Q, S, S, n)
    rng = np.random.default_rng(2)
    g = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    G = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
    d = ctx.degrid2(np.conj(gcf), g, (u, v, None), wb)
    lhs, rhs = np.vdot(g, G), np.vdot(d, vis)
    assert abs(lhs - rhs) / abs(lhs) < 1e-11


def test_device_path_with_torch_tensors(ctx, oracle):
    import torch
    N, W, Q, S, n = 256, 8, 8, 15, 40000
    gcf, u, v, wb, vis = case(55, N, N, W, Q, S, S, n)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    ctx.enable_timing(True)
    ctx.convgrid2(t(gcf), G, (t(u), t(v), None), t(wb), t(vis))
    total, pre, ker = ctx.last_timing()
    ctx.enable_timing(False)
    torch.cuda.synchronize()
    assert rel(G.cpu().numpy(), ref) < TOL
    assert total > 0 and ker > 0
    d = ctx.degrid2(t(gcf), G, (t(u), t(v), None), t(wb))
    torch.cuda.synchronize()
    assert rel(d.cpu().numpy(), oracle.degrid2(gcf, ref, u, v, wb)) < 1e-9


def test_device_path_is_ordered_with_the_callers_stream(ctx):
    """Regression: torch's current stream is HIP's null stream (pointer 0).  The device-pointer calls
    must be enqueued THERE, behind the kernels that are still producing their inputs; a private
    stream would read half-generated coordinates on the first call."""
    import torch
    dev = torch.device("cuda:0")
    N, W, Q, S, n = 1024, 16, 8, 15, 20_000_000
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    gcf = torch.complex(torch.randn((W, Q, Q, S, S), generator=gen, device=dev, dtype=torch.float64),
                        torch.randn((W, Q, Q, S, S), generator=gen, device=dev, dtype=torch.float64))
    torch.cuda.synchronize()
    # a long chain of asynchronous producers, then the gridder immediately behind them
    u = (torch.rand(n, generator=gen, device=dev, dtype=torch.float64) - 0.5) * 0.9
    v = (torch.rand(n, generator=gen, device=dev, dtype=torch.float64) - 0.5) * 0.9
    for _ in range(3):
        u = torch.sin(u) * 1.0001
        v = torch.sin(v) * 1.0001
    wb = torch.randint(0, W, (n,), generator=gen, device=dev, dtype=torch.int64)
    vis = torch.complex(torch.randn(n, generator=gen, device=dev, dtype=torch.float64),
                        torch.randn(n, generator=gen, device=dev, dtype=torch.float64))
    G1 = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    ctx.convgrid2(gcf, G1, (u, v, None), wb, vis)      # no synchronisation in between
    s1 = G1.sum()                                       # consumer on torch's stream, again unsynchronised
    torch.cuda.synchronize()
    G2 = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    ctx.convgrid2(gcf, G2, (u, v, None), wb, vis)
    torch.cuda.synchronize()
    assert ctx.get_option("errors") == 0
    assert ((G1 - G2).abs().max() / G2.abs().max()).item() < 1e-12
    assert abs((s1 - G2.sum()).item()) / G2.abs().sum().item() < 1e-12


@pytest.mark.parametrize("prepass,S", [(1, 7), (2, 15), (2, 21)])
def test_device_path_can_be_captured_into_a_hip_graph(ctx, prepass, S):
    """After a warm-up call (scratch sized, LDS limits raised) a device-pointer gridding call enqueues nothing but
    kernel launches on the caller's stream - no hipMemsetAsync (memset nodes captured from small, odd-sized memsets
    did not replay correctly on ROCm 7.2), no allocation, no synchronisation - so the caller can capture it and replay
    it.  Three replays, each onto a cleared grid, must reproduce the eager result (every replay re-clears and refills
    the pre-pass tables)."""
    import torch
    dev = torch.device("cuda:0")
    N, W, Q, n = 320, 8, 4, 150000
    gcf, u, v, wb, vis = case(91 + S, N, N, W, Q, S, S, n, spread=0.55)
    t = lambda a: torch.from_numpy(a).to(dev)
    tg, tu, tv, twb, tvis = t(gcf), t(u), t(v), t(wb), t(vis)
    try:
        ctx.set_option("prepass", prepass)
        ctx.set_option("sort", 1)
        ref = torch.zeros((N, N), dtype=torch.complex128, device=dev)
        ctx.convgrid2(tg, ref, (tu, tv, None), twb, tvis)
        torch.cuda.synchronize()
        G = torch.zeros_like(ref)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ctx.convgrid2(tg, G, (tu, tv, None), twb, tvis)  # warm-up on the capture stream
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            ctx.convgrid2(tg, G, (tu, tv, None), twb, tvis)
        torch.cuda.synchronize()
        errs = []
        for _ in range(3):
            G.zero_()
            graph.replay()
            torch.cuda.synchronize()
            errs.append(((G - ref).abs().max() / ref.abs().max()).item())
        errors = ctx.get_option("errors")
    finally:
        ctx.set_option("prepass", 0)
        ctx.set_option("sort", 0)
    assert max(errs) < 1e-12 and errors == 0


@pytest.mark.parametrize("mode,n", [(1, 20000), (2, 200000), (4, 200000), (5, 200000), (6, 200000), (3, 20000)])
def test_record_writes_are_bounded_by_the_array(ctx, oracle, mode, n):
    """Every record store of the pre-pass is checked against the record array's capacity.  The test hook
    "fault_inject" hides the last k slots from the scatter (a deliberately short table): nothing is written beyond
    the shortened array, the rejected records are counted in "errors", the host-pointer call reports the
    inconsistency as GRIDHIP_EINVAL instead of returning a wrong grid silently - and the GPU does not fault."""
    import gridhip
    import torch
    N, W, Q, S = 256, 8, 4, 9
    gcf, u, v, wb, vis = case(77, N, N, W, Q, S, S, n, spread=0.45)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    hide = 1000
    try:
        ctx.set_option("prepass", mode)
        ctx.set_option("fault_inject", hide)
        G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
        ctx.convgrid2(t(gcf), G, (t(u), t(v), None), t(wb), t(vis))
        torch.cuda.synchronize()
        errors = ctx.get_option("errors")
        with pytest.raises(gridhip.GridHipError) as ei:
            ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
        ctx.set_option("fault_inject", 0)
        got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
        clean = ctx.get_option("errors")
    finally:
        ctx.set_option("prepass", 0)
        ctx.set_option("fault_inject", 0)
    assert errors >= hide             # (the sorted kernel may count the holes it finds as well)
    assert ei.value.code == gridhip._lib.EINVAL
    assert clean == 0 and rel(got, ref) < TOL
    # the faulty run lost records, it did not scribble: its grid is the reference minus some footprints
    assert np.isfinite(G.cpu().numpy()).all()


@pytest.mark.parametrize("N,W,Q,S,n", [(512, 32, 8, 15, 200000), (200, 4, 2, 9, 3000), (128, 2, 2, 6, 4000)])
def test_plan_bins_once_grids_and_degrids_many_times(ctx, oracle, N, W, Q, S, n):
    """gridhip_plan_*: the records are independent of the visibility and kernel values."""
    import torch
    dev = torch.device("cuda:0")
    gcf, u, v, wb, vis = case(N + n, N, N, W, Q, S, S, n)
    t = lambda a: torch.from_numpy(a).to(dev)
    tg, tu, tv, twb = t(gcf), t(u), t(v), t(wb)
    plan = ctx.plan((N, N), gcf.shape, (tu, tv, None), twb)
    del tu, tv, twb  # the coordinates are no longer needed
    vis2 = vis[::-1].copy() * (0.5 - 2j)
    gcf2 = np.conj(gcf) * 1.5
    for k, (kk, vv) in enumerate([(gcf, vis), (gcf, vis2), (gcf2, vis)]):
        G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
        plan.grid(t(kk), G, t(vv))
        torch.cuda.synchronize()
        ref = oracle.convgrid2(kk, np.zeros((N, N), dtype=np.complex128), u, v, wb, vv)
        assert rel(G.cpu().numpy(), ref) < TOL, k
    rng = np.random.default_rng(3)
    Gd = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    d = plan.degrid(tg, t(Gd))
    torch.cuda.synchronize()
    assert rel(d.cpu().numpy(), oracle.degrid2(gcf, Gd, u, v, wb)) < TOL
    plan.close()


def test_baseline_config2_full_size(ctx, oracle):
    """BASELINE.json configs[1]: 10^6 vis, 2048^2 grid, 7x7 support — small enough to compare outright."""
    N, W, Q, S, n = 2048, 16, 8, 7, 1_000_000
    gcf, u, v, wb, vis = case(2026, N, N, W, Q, S, S, n, spread=0.49)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
    assert rel(got, ref) < TOL


def test_bench_workload_against_the_oracle_outright(ctx, oracle):
    """BASELINE.json configs[2] - the very stream and kernel table bench.py times (10^8 visibilities, 4096^2, 128
    planes, 15x15, Q = 8; bench.synth_vis / synth_kernels with the bench's seed) - gridded by the GPU and by the CPU
    oracle in its owner-computes mode (bit-identical to the serial oracle, tests/test_oracle.py), compared cell by
    cell.  The oracle runs at ~16 Mvis/s on the GPU box's 128 host threads; with fewer than 32 the stream is cut to
    2 x 10^7 visibilities so that the test stays within a minute."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda:0")
    n, N, W, Q, S = bench.WORKLOADS["cfg3"]
    if (os.cpu_count() or 1) < 32:
        n = 20_000_000
    gcf = bench.synth_kernels(W, Q, S, dev)
    u, v, wb, vis = bench.synth_vis(n, N, W, S, 0x5EEDC0DE, dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    torch.cuda.synchronize()
    assert ctx.get_option("errors") == 0 and ctx.last_dropped() == 0
    got = G.cpu().numpy()
    d = ctx.degrid2(gcf, G, (u, v, None), wb)       # the adjoint pass over the same stream, checked on a sample
    ref = oracle.convgrid2(gcf.cpu().numpy(), np.zeros((N, N), dtype=np.complex128), u.cpu().numpy(), v.cpu().numpy(),
                           wb.cpu().numpy(), vis.cpu().numpy(), mt_mode=2)
    assert rel(got, ref) < TOL
    k = slice(0, 200_000)
    dref = oracle.degrid2(gcf.cpu().numpy(), ref, u[k].cpu().numpy(), v[k].cpu().numpy(), wb[k].cpu().numpy())
    assert rel(d[k].cpu().numpy(), dref) < TOL


@pytest.mark.parametrize("N,n", [(2048, 4_000_000), (4096, 100_000_000), (8192, 20_000_000)])
def test_checksum_property_large(ctx, N, n):
    """Size-independent checks at BASELINE.json's full sizes (configs[2]: 10^8 vis on 4096^2; the
    8192^2 grid of configs[4]) with 128 planes, 15x15, Q=8: with every tap in range,
    sum(G) == sum_k vis_k * sum_ij K[slice_k] (a checksum of checksums), and linearity in vis holds
    between two runs."""
    import torch
    dev = torch.device("cuda:0")
    W, Q, S = 128, 8, 15
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    m = (S / 2 + 1) / N
    u = (torch.rand(n, generator=gen, device=dev, dtype=torch.float64) - 0.5) * (1 - 2 * m)
    v = (torch.rand(n, generator=gen, device=dev, dtype=torch.float64) - 0.5) * (1 - 2 * m)
    wb = torch.randint(0, W, (n,), generator=gen, device=dev, dtype=torch.int64)
    vis = torch.complex(torch.randn(n, generator=gen, device=dev, dtype=torch.float64),
                        torch.randn(n, generator=gen, device=dev, dtype=torch.float64))
    gcf = torch.complex(torch.randn((W, Q, Q, S, S), generator=gen, device=dev, dtype=torch.float64),
                        torch.randn((W, Q, Q, S, S), generator=gen, device=dev, dtype=torch.float64))
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    torch.cuda.synchronize()
    # slice index per visibility, recomputed with torch in fp64 (same formula as frac_coord)
    def fc(p):
        x = N // 2 + p * N
        fl = torch.floor(x + 0.5 / Q)
        fr = torch.round((x - fl) * Q).clamp(0, Q - 1)
        return fl.long(), fr.long()
    _, xf = fc(u)
    _, yf = fc(v)
    ksum = gcf.sum(dim=(3, 4))[wb, yf, xf]
    expect = (vis * ksum).sum()
    got = G.sum()
    scale = (vis.abs() * gcf.abs().sum(dim=(3, 4))[wb, yf, xf]).sum()
    assert abs((got - expect).item()) / scale.item() < 1e-12
    assert ctx.get_option("errors") == 0
    gmax = G.abs().max().item()
    # linearity: gridding -2*vis onto G must leave -G (accumulate-into), without a second grid buffer
    ctx.convgrid2(gcf, G, (u, v, None), wb, -2.0 * vis)
    torch.cuda.synchronize()
    assert abs((G.sum() + expect).item()) / scale.item() < 1e-12
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    torch.cuda.synchronize()
    assert (G.abs().max().item() / gmax) < 1e-11  # back to zero up to rounding


def test_alternating_streams_share_the_context_scratch_safely(ctx, oracle):
    """A context has one set of scratch buffers.  Calls issued alternately on two streams, with no host
    synchronisation in between, must not overwrite records another stream's kernels are still reading:
    gridhip_set_stream orders the new stream after the old one."""
    import torch
    dev = torch.device("cuda:0")
    N, W, Q, S, n = 512, 8, 4, 9, 400000
    cases = [case(300 + i, N, N, W, Q, S, S, n, spread=0.5) for i in range(2)]
    t = lambda a: torch.from_numpy(a).to(dev)
    dc = [tuple(t(a) for a in c) for c in cases]
    refs = [oracle.convgrid2(c[0], np.zeros((N, N), dtype=np.complex128), c[1], c[2], c[3], c[4]) for c in cases]
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    G = [torch.zeros((N, N), dtype=torch.complex128, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(6):
        for k in (0, 1):
            with torch.cuda.stream(s[k]):
                gcf, u, v, wb, vis = dc[k]
                ctx.convgrid2(gcf, G[k], (u, v, None), wb, vis)
    torch.cuda.synchronize()
    for k in (0, 1):
        err = np.abs(G[k].cpu().numpy() / 6 - refs[k]).max() / np.abs(refs[k]).max()
        assert err < 1e-10
    assert ctx.get_option("errors") == 0


def test_reserved_cus_do_not_change_the_grid(ctx, oracle):
    gcf, u, v, wb, vis = case(77, 384, 384, 8, 4, 15, 15, 300000, spread=0.5)
    ref = oracle.convgrid2(gcf, np.zeros((384, 384), dtype=np.complex128), u, v, wb, vis)
    try:
        for k in (8, 32, 250, 1000):
            ctx.set_option("reserve_cus", k)
            ctx.set_option("sort", 1)
            got = ctx.convgrid2(gcf, np.zeros((384, 384), dtype=np.complex128), (u, v, None), wb, vis)
            assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-10
    finally:
        ctx.set_option("reserve_cus", 0)
        ctx.set_option("sort", 0)


@pytest.mark.parametrize("n,groups", [(1_500_000, 1), (3_000_000, 2), (12_500_000, 4), (60_000_000, 8)])
def test_automatic_w_groups_follow_the_work_per_tile(ctx, n, groups):
    """The number of w-groups a call gets when option "wgroups" does not say follows the work per tile (ctx.hip,
    profiles/r03_wgroups_by_size.txt: at 6 - 25 x 10^6 visibilities on the headline shape four groups beat eight by 5 - 24 %,
    below that two and one do): read-only option "last_wgroups" shows the choice; the grid's checksum is right with
    every one of them."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    _, N, W, Q, S = bench.WORKLOADS["cfg3"]
    dev = torch.device("cuda:0")
    gcf = bench.synth_kernels(W, Q, S, dev)
    u, v, wb, vis = bench.synth_vis(n, N, W, S, 77, dev)
    G = torch.zeros((N, N), dtype=torch.complex128, device=dev)
    ctx.convgrid2(gcf, G, (u, v, None), wb, vis)
    torch.cuda.synchronize()
    assert ctx.get_option("last_wgroups") == groups and ctx.get_option("last_path") == 1 and ctx.get_option("errors") == 0
    expect, scale = bench.expected_checksum(u, v, wb, vis, gcf, N)
    assert abs(G.sum().item() - expect.item()) / scale.item() < 1e-10


@pytest.mark.parametrize("S", [15, 23])
def test_yielding_cus_do_not_change_the_grid(ctx, oracle, S):
    """Option "yield_cus": part of the tile kernel's work-groups take eight work items and leave, further ones are
    started in their place (so that a kernel queued on another stream gets CUs).  Every work item is still processed
    exactly once: grid and degrid against the oracle for few, many and more-than-there-are CUs."""
    N, n = 1024, 2_000_000
    gcf, u, v, wb, vis = case(500 + S, N, N, 8, 4, S, S, n, spread=0.5)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    rng = np.random.default_rng(S)
    G = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    dref = oracle.degrid2(gcf, G, u, v, wb)
    try:
        for k in (32, 64, 248, 1000):
            ctx.set_option("yield_cus", k)
            ctx.set_option("sort", 1)
            got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
            assert ctx.get_option("last_path") == 1 and ctx.get_option("errors") == 0
            assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-10
            dg = ctx.degrid2(gcf, G, (u, v, None), wb)
            assert np.abs(dg - dref).max() / np.abs(dref).max() < 1e-10
    finally:
        ctx.set_option("yield_cus", 0)
        ctx.set_option("sort", 0)


@pytest.mark.parametrize("S", [15, 21, 9])
def test_bigtile_matches_oracle(ctx, oracle, S):
    """Option "bigtile": the tap-reusing kernel's tile uses all of the LDS, its im plane at a run-time distance from
    the re plane (the BT instantiations).  Only grids with room for 1 024 such tiles get them, hence the grid size;
    grid and degrid against the oracle, and the geometry really is the big one (fewer, larger bins: the pre-pass's
    record count per bin shows in nothing the ABI exposes, so the check is that both settings agree with the oracle
    while their timings differ is left to tools/sweep.py)."""
    N, W, Q, n = 3200, 16, 4, 1_500_000
    gcf, u, v, wb, vis = case(4000 + S, N, N, W, Q, S, S, n, spread=0.52)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    rng = np.random.default_rng(S)
    G = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    dref = oracle.degrid2(gcf, G, u, v, wb)
    try:
        for bt in (1, 2):
            ctx.set_option("bigtile", bt)
            ctx.set_option("sort", 1)
            got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
            assert ctx.get_option("last_path") == 1 and ctx.get_option("errors") == 0
            assert rel(got, ref) < TOL, bt
            d = ctx.degrid2(gcf, G, (u, v, None), wb)
            assert rel(d, dref) < TOL, bt
    finally:
        ctx.set_option("bigtile", 0)
        ctx.set_option("sort", 0)


def test_last_path_reports_which_gridder_ran(ctx):
    """read-only option "last_path": 1 = tap-reusing kernel, 2 = through sub-footprints, 3 = general tile kernel,
    4 = direct atomics - a call that falls off the fast path says so (include/gridhip.h, "Limits")"""
    def run(S, gw, n, **opts):
        gcf, u, v, wb, vis = case(1, 256, 256, 2, 2, S, gw, n)
        try:
            for k, val in opts.items():
                ctx.set_option(k, val)
            ctx.convgrid2(gcf, np.zeros((256, 256), dtype=np.complex128), (u, v, None), wb, vis)
            return ctx.get_option("last_path")
        finally:
            for k in opts:
                ctx.set_option(k, 0)
    assert run(15, 15, 60000, sort=1) == 1
    assert run(25, 25, 60000, sort=1) == 1          # parts of the tap list: still one record per visibility
    assert run(25, 25, 60000, sort=1, subfoot=1) == 2
    assert run(9, 5, 60000, sort=1) == 2             # non-square: sub-footprints
    assert run(15, 15, 500) == 3                     # too few visibilities per work item for the sort to pay
    assert run(15, 15, 60000, sort=2) == 3
    assert run(15, 15, 60000, variant=1) == 4


@pytest.mark.parametrize("opts", [{"wgroups": 4}, {"wgroups": 1}, {"wgroups": 16}, {"wgroups": 2, "subfoot": 1}])
def test_bigtile_keeps_the_fast_path_whatever_the_wgroups(ctx, oracle, opts):
    """The big tile is sized around the sort's histogram, whose length depends on the number of w-groups (and, for
    sub-footprints, parts): a geometry whose histogram would not fit beside the big planes must fall back to the classic
    tile, not off the tap-reusing kernel (round 3 found wgroups = 4 on the 8192^2 grid taking the general kernel)."""
    N, W, Q, S, n = 3200, 64, 8, 17, 600_000
    gcf, u, v, wb, vis = case(77 + len(opts), N, N, W, Q, S, S, n, spread=0.5)
    ref = oracle.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), u, v, wb, vis, mt_mode=2)
    try:
        ctx.set_option("bigtile", 1)
        ctx.set_option("sort", 1)
        for k, val in opts.items():
            ctx.set_option(k, val)
        got = ctx.convgrid2(gcf, np.zeros((N, N), dtype=np.complex128), (u, v, None), wb, vis)
        path = ctx.get_option("last_path")
    finally:
        for k in ("bigtile", "sort", "wgroups", "subfoot"):
            ctx.set_option(k, 0)
    assert path in (1, 2) and rel(got, ref) < TOL
