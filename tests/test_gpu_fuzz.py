"""Seeded fuzz: random grid shapes, kernel shapes, oversampling, plane counts, distributions and
tuning options; convgrid2 and degrid2 through the C ABI against the CPU oracle (1e-10 relative).
Catches shape-dependent indexing mistakes (LDS pitch, tile offsets, halo clipping, chunk/batch
boundaries) that the fixed-shape parity tests could miss."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make_case(seed, large=False):
    rng = np.random.default_rng(1000 + seed)
    H = int(rng.integers(17, 400))
    Wd = H if rng.random() < 0.5 else int(rng.integers(17, 400))
    square = rng.random() < (0.85 if large else 0.7)
    if large:  # supports 17 .. 32: the tap-reusing kernel takes a slice's taps in parts; others through sub-footprints
        gh = int(rng.integers(17, 33))
        gw = gh if square else int(rng.choice([18, 23, 27, 36, 40]))
    else:
        gh = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 11, 13, 15, 17, 21, 25, 31, 33]))
        gw = gh if square else int(rng.choice([1, 3, 4, 5, 7, 9, 12, 15, 19, 28]))
    Q = int(rng.choice([1, 2, 3, 4, 8]))
    W = int(rng.choice([1, 2, 5, 8, 13, 32]))
    n = int(rng.choice([1, 7, 300, 5000, 40000, 120000]))
    spread = float(rng.choice([0.3, 0.5, 0.56]))
    if rng.random() < 0.3:
        u = np.clip(rng.normal(0, 0.05, n), -0.7, 0.7)
        v = np.clip(rng.normal(0, 0.05, n), -0.7, 0.7)
    else:
        u, v = rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n)
    gcf = rng.normal(size=(W, Q, Q, gh, gw)) + 1j * rng.normal(size=(W, Q, Q, gh, gw))
    wb = rng.integers(0, W, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    opts = {}
    if rng.random() < 0.6:
        opts["tile"] = int(rng.choice([8, 16, 32, 64]))
    if rng.random() < 0.6:
        opts["block"] = int(rng.choice([64, 128, 256, 512, 1024]))
    if rng.random() < 0.5:
        opts["wgroups"] = int(rng.choice([1, 2, 3, 8, 16]))
    if rng.random() < 0.5:
        opts["chunk"] = int(rng.choice([64, 100, 1000, 5000]))
    opts["sort"] = int(rng.choice([0, 1, 2]))
    opts["prepass"] = int(rng.choice([0, 1, 2, 4, 5, 6]))  # one- or two-level scatter in the binning pre-pass (and its variants)
    if large:
        opts["sort"] = int(rng.choice([0, 1, 1]))
        opts["bigtile"] = int(rng.choice([0, 1, 2]))
        opts["subfoot"] = int(rng.choice([0, 0, 1]))
        opts["reserve_cus"] = int(rng.choice([0, 8, 32, 200]))
        if "tile" in opts and rng.random() < 0.5:
            del opts["tile"]  # (bigtile only applies to automatically chosen tiles)
    opts["yield_cus"] = int(rng.choice([0, 0, 32, 64, 250]))  # (ignored while reserve_cus is set)
    return (H, Wd, gcf, u, v, wb, vis, opts)


# GRIDHIP_FUZZ_EXTRA=k adds k more seeds of every kind (a one-off wider run; the committed suite stays at minutes)
_EXTRA = int(os.environ.get("GRIDHIP_FUZZ_EXTRA", "0"))


@pytest.mark.parametrize("seed", list(range(40)) + [f"L{i}" for i in range(24)] + list(range(1000, 1000 + _EXTRA)) +
                         [f"L{i}" for i in range(100, 100 + _EXTRA)])
def test_fuzz_convgrid2_and_degrid2(ctx, oracle, seed):
    large = isinstance(seed, str)
    seed = 500 + int(seed[1:]) if large else seed
    H, Wd, gcf, u, v, wb, vis, opts = make_case(seed, large)
    keys = ("tile", "block", "wgroups", "chunk", "sort", "prepass", "bigtile", "subfoot", "reserve_cus", "yield_cus")
    G0 = np.zeros((H, Wd), dtype=np.complex128)
    ref = oracle.convgrid2(gcf, G0.copy(), u, v, wb, vis)
    rng = np.random.default_rng(seed)
    Gd = rng.normal(size=(H, Wd)) + 1j * rng.normal(size=(H, Wd))
    dref = oracle.degrid2(gcf, Gd, u, v, wb)
    try:
        try:
            for k, val in opts.items():
                ctx.set_option(k, val)
            got = ctx.convgrid2(gcf, G0.copy(), (u, v, None), wb, vis)
            dgot = ctx.degrid2(gcf, Gd, (u, v, None), wb)
        except Exception as e:  # a forced tile that cannot fit its halo in LDS is a legal refusal
            if "LDS" in str(e) or "unsupported" in str(e).lower():
                pytest.skip(f"shape refused with these options: {e}")
            raise
    finally:
        for k in keys:
            ctx.set_option(k, 0)
    assert ctx.get_option("errors") == 0
    assert rel(got, ref) < TOL, (H, Wd, gcf.shape, len(u), opts)
    if np.abs(dref).max() > 0:
        assert rel(dgot, dref) < TOL, (H, Wd, gcf.shape, len(u), opts)


@pytest.mark.parametrize("seed", list(range(16)) + list(range(100, 100 + _EXTRA)))
def test_fuzz_awgrid(ctx, oracle, seed):
    """aw gridders: random supports (compile-time and generic build kernels, tap-reusing and general tile kernels),
    antenna / plane / oversampling counts, repeated and unique keys, the per-key cache on and off, bad indices."""
    rng = np.random.default_rng(7000 + seed)
    S = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 8, 12, 16, 19]))
    N = int(rng.integers(2 * S + 4, 260))
    W, Q, A = int(rng.choice([1, 2, 5])), int(rng.choice([1, 2, 4])), int(rng.choice([2, 3, 9]))
    nb = int(rng.choice([1, 5, 60, 400]))
    dumps = int(rng.choice([1, 3, 8]))
    n = nb * dumps
    wk = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
    ak = rng.normal(size=(A, S, S)) + 1j * rng.normal(size=(A, S, S))
    u0, v0 = rng.uniform(-0.55, 0.55, nb), rng.uniform(-0.55, 0.55, nb)
    drift = float(rng.choice([0.0, 0.02, 0.3])) / N
    d = np.arange(dumps)
    u = (u0[:, None] + d[None, :] * drift).ravel()
    v = (v0[:, None] - d[None, :] * drift).ravel()
    rep = lambda a: np.repeat(a, dumps)
    wb, a1, a2 = rep(rng.integers(0, W, nb)), rep(rng.integers(0, A, nb)), rep(rng.integers(0, A, nb))
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    if n > 4:
        wb[1], a2[3] = W + 2, -5
    keep = (wb >= 0) & (wb < W) & (a2 >= 0)
    start = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    ref = oracle.awgrid(wk, ak, start.copy(), u[keep], v[keep], wb[keep], a1[keep], a2[keep], vis[keep], direct=True)
    cache = int(rng.integers(0, 2))
    sort = int(rng.choice([0, 2]))
    try:
        ctx.set_option("aw_cache", cache)
        ctx.set_option("sort", sort)
        got = ctx.convgrid4(wk, ak, start.copy(), (u, v, None), (wb, a1, a2), vis)
        st = ctx.aw_stats(S)
    finally:
        ctx.set_option("aw_cache", 1)
        ctx.set_option("sort", 0)
    assert ctx.get_option("errors") == 0
    assert rel(got, ref) < TOL, (S, N, W, Q, A, nb, dumps, cache, sort)
    assert st["vis_keyed"] == n and (st["kernels_built"] == n if not cache else st["kernels_built"] <= max(int(keep.sum()), 1))
