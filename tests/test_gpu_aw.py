"""GPU parity for the AW-projection gridders (convgrid3 / convgrid4, aw_kernel_fn2, convolve2d with its
transposing pad, aw_imaging) against the CPU oracle, which evaluates convolve2d the way the reference
does (FFT path).  The GPU evaluates the algebraically identical direct form, so agreement is a
tolerance (observed ~1e-14); "parity unpinned" by the reference itself."""
import numpy as np
import pytest

from oracle import gridref_np as P

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_smalltest_aw_fixture(ctx, golden):
    """The literal inputs of the reference's test/SmallTest.hs:51-76 (15x15 kernels on a 10x10 grid)."""
    g = golden("smalltest_aw")
    G = ctx.convgrid4(g["wkerns"], g["akerns"], np.zeros((10, 10), dtype=np.complex128), (g["u"], g["v"], g["w"]),
                      (g["wbin"], g["a1"], g["a2"]), g["vis"])
    assert rel(G, g["expected"]) < TOL
    G3 = ctx.convgrid3(g["wkerns"], g["akerns"], np.zeros((10, 10), dtype=np.complex128), (g["u"], g["v"], g["w"]),
                       (g["wbin"], g["a1"], g["a2"]), g["vis"])
    assert np.abs(G3 - G).max() <= 1e-15 * np.abs(G).max()


@pytest.mark.parametrize("N,W,Q,S,A,n", [(64, 3, 2, 15, 6, 300), (96, 2, 4, 7, 12, 500), (80, 4, 1, 9, 3, 200),
                                         (72, 2, 2, 16, 4, 150), (90, 2, 2, 19, 3, 60), (64, 2, 2, 4, 5, 300),
                                         (64, 2, 2, 5, 4, 300), (72, 3, 2, 11, 5, 250), (80, 2, 3, 13, 4, 250)])
def test_awgrid_matches_oracle(ctx, oracle, N, W, Q, S, A, n):
    """every compile-time support of the kernel builder (5 .. 15) and the generic builder (4, 16, 19)"""
    rng = np.random.default_rng(N + S)
    wk = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
    ak = rng.normal(size=(A, S, S)) + 1j * rng.normal(size=(A, S, S))
    u, v = rng.uniform(-0.55, 0.55, n), rng.uniform(-0.55, 0.55, n)
    wb, a1, a2 = rng.integers(0, W, n), rng.integers(0, A, n), rng.integers(0, A, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    ref = oracle.awgrid(wk, ak, np.zeros((N, N), dtype=np.complex128), u, v, wb, a1, a2, vis)  # FFT convolve2d
    got = ctx.convgrid4(wk, ak, np.zeros((N, N), dtype=np.complex128), (u, v, None), (wb, a1, a2), vis)
    assert rel(got, ref) < TOL
    # accumulate-into + bad indices contribute nothing
    bad = (np.array([0, W, 0]), np.array([0, 0, A]), np.array([0, 1, -1]))
    G0 = ref.copy()
    out = ctx.convgrid4(wk, ak, G0, (u[:3], v[:3], None), bad, vis[:3])
    one = oracle.awgrid(wk, ak, ref.copy(), u[:1], v[:1], bad[0][:1], bad[1][:1], bad[2][:1], vis[:1])
    assert rel(out, one) < TOL


def test_aw_kernel_is_transposed_same_conv(ctx, oracle):
    """aw_kernel_fn2 (:761-775) through the GPU path: a delta visibility at the grid centre leaves
    conj(awkern) on the grid; compare with the oracle's FFT-path aw_kernel_fn2."""
    rng = np.random.default_rng(3)
    S, Q, N = 15, 2, 32
    wk = rng.normal(size=(1, Q, Q, S, S)) + 1j * rng.normal(size=(1, Q, Q, S, S))
    ak = rng.normal(size=(2, S, S)) + 1j * rng.normal(size=(2, S, S))
    G = ctx.convgrid4(wk, ak, np.zeros((N, N), dtype=np.complex128), (np.array([0.0]), np.array([0.0]), None),
                      (np.array([0]), np.array([0]), np.array([1])), np.array([1 + 0j]))
    x0 = N // 2 - S // 2
    got = G[x0:x0 + S, x0:x0 + S]
    ref = np.conj(oracle.aw_kernel_fn2(0, 0, wk[0], ak[0], ak[1]))
    assert rel(got, ref) < TOL
    assert rel(got, np.conj(P.same_conv_direct(P.same_conv_direct(ak[0], ak[1]), wk[0, 0, 0].T))) < TOL


def test_aw_imaging(ctx, oracle):
    rng = np.random.default_rng(8)
    theta, lam = 0.05, 1280  # N = 64
    W, Q, S, A, n = 5, 2, 15, 4, 200
    wk = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
    ak = rng.normal(size=(A, S, S)) + 1j * rng.normal(size=(A, S, S))
    wvals = np.sort(rng.uniform(-400, 400, W))
    u, v, w = rng.uniform(-600, 600, n), rng.uniform(-600, 600, n), rng.uniform(-500, 500, n)
    a1, a2 = rng.integers(0, A, n), rng.integers(0, A, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    got = ctx.aw_imaging(theta, lam, wk, wvals, ak, (u, v, w), (a1, a2, None, None), vis)
    wb = np.array([oracle.find_closest(wvals, x) for x in w])
    ref = oracle.awgrid(wk, ak, np.zeros((64, 64), dtype=np.complex128), u / lam, v / lam, wb, a1, a2, vis)
    assert rel(got, ref) < TOL


def _aw_case(seed, N, W, Q, S, A, nb, dumps, drift=0.02):
    """baseline-structured stream: nb baselines x `dumps` consecutive samples drifting by `drift` cells each"""
    rng = np.random.default_rng(seed)
    wk = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
    ak = rng.normal(size=(A, S, S)) + 1j * rng.normal(size=(A, S, S))
    u0, v0 = rng.uniform(-0.4, 0.4, nb), rng.uniform(-0.4, 0.4, nb)
    ang = rng.uniform(0, 2 * np.pi, nb)
    d = np.arange(dumps)
    u = (u0[:, None] + d[None, :] * np.cos(ang)[:, None] * drift / N).ravel()
    v = (v0[:, None] + d[None, :] * np.sin(ang)[:, None] * drift / N).ravel()
    rep = lambda a: np.repeat(a, dumps)
    wb, a1, a2 = rep(rng.integers(0, W, nb)), rep(rng.integers(0, A, nb)), rep(rng.integers(0, A, nb))
    n = nb * dumps
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    return wk, ak, u, v, wb, a1, a2, vis


@pytest.mark.parametrize("S", [15, 7, 9, 12])
def test_aw_key_cache_on_and_off(ctx, oracle, S):
    """Per-key de-duplication (option aw_cache): the kernel of every distinct (a1, a2, wbin, yf, xf) is built once and
    the visibilities that share it reuse it.  Same grid with the cache on and off (where every visibility gets a
    private kernel, as the reference evaluates), both against the oracle; the stats report the repetition."""
    N, W, Q, A = 192, 3, 4, 5
    wk, ak, u, v, wb, a1, a2, vis = _aw_case(40 + S, N, W, Q, S, A, 400, 6)
    wb[::37] = W      # out of range: dropped and counted, with the cache on and off
    a1[5::41] = -1
    keep = (wb < W) & (a1 >= 0)
    ref = oracle.awgrid(wk, ak, np.zeros((N, N), dtype=np.complex128), u[keep], v[keep], wb[keep], a1[keep], a2[keep],
                        vis[keep], direct=True)
    try:
        ctx.set_option("aw_cache", 1)
        on = ctx.convgrid4(wk, ak, np.zeros((N, N), dtype=np.complex128), (u, v, None), (wb, a1, a2), vis)
        st_on, drop_on, err_on = ctx.aw_stats(S), ctx.last_dropped(), ctx.get_option("errors")
        ctx.set_option("aw_cache", 0)
        off = ctx.convgrid4(wk, ak, np.zeros((N, N), dtype=np.complex128), (u, v, None), (wb, a1, a2), vis)
        st_off, drop_off = ctx.aw_stats(S), ctx.last_dropped()
    finally:
        ctx.set_option("aw_cache", 1)
    assert rel(on, ref) < TOL and rel(off, ref) < TOL and err_on == 0
    assert drop_on == drop_off == int((~keep).sum())
    assert st_off["kernels_built"] == st_off["vis_keyed"] == len(u)
    assert st_on["vis_keyed"] == len(u) and st_on["kernels_built"] < 0.6 * int(keep.sum())
    assert st_on["hit_rate"] > 0.4


def test_aw_at_the_configured_grid_size(ctx, oracle):
    """BASELINE configs[3]'s grid (4096^2, 15x15, 128 planes, 512 antennas): a sample small enough for the oracle is
    checked through the checksum sum(G) = sum_k vis_k * sum_ij conj(awkern_k)[i, j] (every tap lands inside the grid),
    and a stream of 5 x 10^6 visibilities (two batches of the kernel table) through linearity in the visibilities."""
    import torch
    N, W, Q, S, A = 4096, 128, 8, 15, 512
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # (smooth kernels keep the checksum well conditioned)
    rng = np.random.default_rng(2)
    j = np.arange(S) - S // 2
    base = np.exp(-(j[:, None] ** 2 + j[None, :] ** 2) / 18.0)
    wk = base[None, None, None] * np.exp(1j * rng.uniform(0, 0.5, size=(W, Q, Q, 1, 1))) * (1 + 0.1 * rng.normal(size=(W, Q, Q, S, S)))
    ak = base[None] * np.exp(1j * rng.uniform(0, 0.5, size=(A, 1, 1))) * (1 + 0.1 * rng.normal(size=(A, S, S)))
    nb, dumps = 2500, 8
    _, _, u, v, wb, a1, a2, vis = _aw_case(11, N, W, Q, S, A, nb, dumps)
    G = ctx.convgrid4(t(wk), t(ak), torch.zeros((N, N), dtype=torch.complex128, device=dev), (t(u), t(v), None),
                      (t(wb), t(a1), t(a2)), t(vis))
    st = ctx.aw_stats(S)
    total = complex(G.sum().item())
    # the oracle's kernels, one per DISTINCT key of the sample
    keys = {}
    expect = 0j
    _, xfs = oracle.frac_coord(N, Q, u)
    _, yfs = oracle.frac_coord(N, Q, v)
    for k in range(len(u)):
        key = (int(a1[k]), int(a2[k]), int(wb[k]), int(yfs[k]), int(xfs[k]))
        if key not in keys:
            keys[key] = np.conj(oracle.aw_kernel_fn2(key[3], key[4], wk[key[2]], ak[key[0]], ak[key[1]], direct=True)).sum()
        expect += vis[k] * keys[key]
    assert abs(total - expect) / abs(expect) < 1e-9
    assert st["kernels_built"] == len(keys) and ctx.get_option("errors") == 0 and ctx.last_dropped() == 0
    # linearity at 5 x 10^6 visibilities (> one batch of the kernel table)
    nb = 625_000
    _, _, u, v, wb, a1, a2, vis = _aw_case(12, N, W, Q, S, A, nb, dumps)
    v2 = np.random.default_rng(5).normal(size=len(u)) + 1j * np.random.default_rng(6).normal(size=len(u))
    tu, tv, twb, ta1, ta2, twk, tak = t(u), t(v), t(wb), t(a1), t(a2), t(wk), t(ak)
    z = lambda: torch.zeros((N, N), dtype=torch.complex128, device=dev)
    Ga = ctx.convgrid4(twk, tak, z(), (tu, tv, None), (twb, ta1, ta2), t(vis))
    st = ctx.aw_stats(S)
    Gb = ctx.convgrid4(twk, tak, z(), (tu, tv, None), (twb, ta1, ta2), t(v2))
    Gc = ctx.convgrid4(twk, tak, z(), (tu, tv, None), (twb, ta1, ta2), t(2.0 * vis - 0.5j * v2))
    lin = (Gc - (2.0 * Ga - 0.5j * Gb)).abs().max().item() / Gc.abs().max().item()
    assert lin < 1e-11 and ctx.get_option("errors") == 0
    assert st["vis_keyed"] == len(u) and 0.3 < st["hit_rate"] < 0.95
