"""CPU tests of the oracle itself: the one KAT the reference records, derived KATs from the
literal inputs of the reference's test scripts, C-vs-numpy agreement, and the algebraic
properties the GPU parity tests later rely on.  (No GPU, no /root/reference.)"""
import numpy as np
import pytest

from oracle import gridref_np as P


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def rand_case(rng, N=48, W=3, Q=4, S=7, n=500, spread=0.56, M=None):
    M = M or N
    gcf = rng.normal(size=(W, Q, Q, S, S)) + 1j * rng.normal(size=(W, Q, Q, S, S))
    u = rng.uniform(-spread, spread, n)
    v = rng.uniform(-spread, spread, n)
    wb = rng.integers(0, W, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    return gcf, u, v, wb, vis


# ---- the one output the reference itself records (old/BrokenNumbers.hs:86-91) --------------
def test_kat_brokennumbers_recorded(oracle, golden):
    g = golden("brokennumbers")
    # `permute (+) origin indexer source` applied twice onto a zero 5x5 grid; expressed through
    # grid(): cell = N/2 + floor(.5 + N*p)  =>  p = (cell - 2) / 5
    for impl in (oracle, P):
        G = np.zeros((5, 5), dtype=np.complex128)
        for _ in range(int(g["passes"])):
            impl.grid(G, (g["x"] - 2) / 5.0, (g["y"] - 2) / 5.0, g["val"])
        assert np.array_equal(G, g["expected"])


def test_kat_brokennumbers_real_recorded(oracle, golden):
    """the real-valued twin the reference records too (old/BrokenNumbers.hs:101-106; interpreter and CPU backend agree)"""
    g = golden("brokennumbers_real")
    for impl in (oracle, P):
        G = np.zeros((5, 5), dtype=np.complex128)
        for _ in range(int(g["passes"])):
            impl.grid(G, (g["x"] - 2) / 5.0, (g["y"] - 2) / 5.0, g["val"].astype(np.complex128))
        assert np.array_equal(G.real, g["expected"]) and not G.imag.any()


# ---- derived KATs (reference-literal inputs) -------------------------------------------------
def test_kat_fixbounds_derived(oracle, golden):
    """testFixbounds (test/GridTesting.hs:365-387): ten points scattered with the four offsets of a 2x2 footprint,
    out-of-range ones dropped by fixoutofboundsOLD - convgrid with a 2x2 kernel of ones."""
    g = golden("fixbounds")
    for impl in (oracle, P):
        x, xf = impl.frac_coord(5, 1, g["pu"])
        y, yf = impl.frac_coord(5, 1, g["pv"])
        assert np.array_equal(x - 1, g["x"]) and np.array_equal(y - 1, g["y"]) and not xf.any() and not yf.any()
        G = np.zeros((5, 5), dtype=np.complex128)
        impl.convgrid(g["gcf"], G, g["pu"], g["pv"], g["vis"])
        assert np.array_equal(G, g["expected"])  # small integers: exact


def test_kat_fixbounds2_derived(oracle, golden):
    g = golden("fixbounds2")
    for impl in (oracle, P):
        x, xf = impl.frac_coord(5, 2, g["pu"])
        y, yf = impl.frac_coord(5, 2, g["pv"])
        assert np.array_equal(x - 1, g["x"]) and np.array_equal(xf, g["xf"])
        assert np.array_equal(y - 1, g["y"]) and np.array_equal(yf, g["yf"])
        G = np.zeros((5, 5), dtype=np.complex128)
        impl.convgrid(g["gcf"], G, g["pu"], g["pv"], g["vis"])
        assert np.array_equal(G, g["expected"])  # small integers: exact


def test_kat_smalltest_aw_derived(oracle, golden):
    g = golden("smalltest_aw")
    for direct in (False, True):
        G = np.zeros((10, 10), dtype=np.complex128)
        oracle.awgrid(g["wkerns"], g["akerns"], G, g["u"], g["v"], g["wbin"], g["a1"], g["a2"], g["vis"],
                      direct=direct)
        assert rel(G, g["expected"]) < 1e-12
    assert np.abs(g["expected"]).max() > 0


def test_golden_convgrid2_small(oracle, golden):
    g = golden("convgrid2_small")
    N = g["expected"].shape[0]
    G = np.zeros((N, N), dtype=np.complex128)
    oracle.convgrid2(g["gcf"], G, g["u"], g["v"], g["wbin"], g["vis"])
    assert rel(G, g["expected"]) < 1e-14
    d = oracle.degrid2(g["gcf"], g["expected"], g["u"], g["v"], g["wbin"])
    assert rel(d, g["degrid"]) < 1e-14


@pytest.mark.parametrize("w", [100, 1000])
def test_golden_wkernel(oracle, golden, w):
    g = golden(f"wkernel_w{w}")
    k = oracle.w_kernel(float(g["theta"]), float(g["w"]), int(g["npixFF"]), int(g["npixKern"]), int(g["qpx"]))
    assert rel(k, g["expected"]) < 1e-12


# ---- C restatement vs numpy restatement ---------------------------------------------------------
@pytest.mark.parametrize("N,M,W,Q,S,n", [(48, 48, 3, 4, 7, 600), (40, 56, 2, 8, 15, 300), (33, 33, 1, 1, 1, 200),
                                         (64, 64, 4, 3, 5, 500)])
def test_c_vs_numpy_convgrid2(oracle, N, M, W, Q, S, n):
    rng = np.random.default_rng(N * 1000 + S)
    gcf, u, v, wb, vis = rand_case(rng, N, W, Q, S, n)
    G1 = np.zeros((N, M), dtype=np.complex128)
    G2 = np.zeros((N, M), dtype=np.complex128)
    oracle.convgrid2(gcf, G1, u, v, wb, vis)
    P.convgrid2(gcf, G2, u, v, wb, vis)
    assert rel(G1, G2) < 1e-13
    assert rel(oracle.degrid2(gcf, G1, u, v, wb), P.degrid2(gcf, G1, u, v, wb)) < 1e-13
    for mode in (0, 1):
        G3 = np.zeros((N, M), dtype=np.complex128)
        oracle.convgrid2(gcf, G3, u, v, wb, vis, mt_mode=mode, nthreads=4)
        assert rel(G3, G1) < 1e-12


def test_c_vs_numpy_frac_coord(oracle):
    rng = np.random.default_rng(7)
    p = rng.uniform(-0.7, 0.7, 20000)
    for n, q in [(2048, 8), (4096, 8), (5, 2), (1801, 3), (2400, 1)]:
        a, b = oracle.frac_coord(n, q, p)
        c, d = P.frac_coord(n, q, p)
        assert np.array_equal(a, c) and np.array_equal(b, d)
        assert b.min() >= 0 and b.max() <= q - 1


def test_c_vs_numpy_helpers(oracle):
    rng = np.random.default_rng(11)
    w = rng.uniform(-9000, 21000, 1000)
    wb1, mn1, np1 = oracle.wbins(w, 2000)
    wb2, mn2, np2 = P.wbins(w, 2000)
    assert np.array_equal(wb1, wb2) and (mn1, np1) == (mn2, np2)
    assert wb1.min() == 0 and wb1.max() == np1 - 1
    ws = np.sort(rng.uniform(0, 100, 33))
    for x in rng.uniform(-10, 120, 300):
        assert oracle.find_closest(ws, x) == P.find_closest(ws, x)
    for x in ws:  # exact hits find themselves or an equal-distance neighbour
        assert abs(ws[oracle.find_closest(ws, x)] - x) <= np.abs(ws - x).min() + 1e-12
    N = 64
    u, v = rng.uniform(-0.5, 0.5, 500), rng.uniform(-0.5, 0.5, 500)
    vis = rng.normal(size=500) + 1j * rng.normal(size=500)
    assert rel(oracle.doweight(N, u, v, vis), P.doweight(N, u, v, vis)) < 1e-15
    mu = oracle.mirror_uvw(u, v, w[:500], vis)
    nu = P.mirror_uvw(u, v, w[:500], vis)
    for a, b in zip(mu, nu):
        assert np.array_equal(a, b)
    assert (mu[1] >= 0).all()
    for n in (8, 9):
        g = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        assert np.array_equal(oracle.make_grid_hermitian(g), P.make_grid_hermitian(g))


def test_c_vs_numpy_fft_and_kernels(oracle):
    rng = np.random.default_rng(13)
    for n in (8, 15, 30, 64):
        a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        assert rel(oracle.fft2_centered(a, True), P.ifft_c(a)) < 1e-13
        assert rel(oracle.fft2_centered(a, False), P.fft_c(a)) < 1e-13
    a1 = rng.normal(size=(15, 15)) + 1j * rng.normal(size=(15, 15))
    a2 = rng.normal(size=(15, 15)) + 1j * rng.normal(size=(15, 15))
    c_fft = oracle.convolve2d(a1, a2)
    assert rel(c_fft, P.convolve2d(a1, a2)) < 1e-13
    # the padder transpose quirk (src/Gridding.hs:875): convolve2d == same_conv^T
    assert rel(c_fft, P.same_conv_direct(a1, a2).T) < 1e-12
    assert rel(oracle.convolve2d(a1, a2, direct=True), c_fft) < 1e-12
    assert rel(c_fft, P.same_conv_direct(a1, a2)) > 1e-3  # and it is NOT the untransposed one
    for (nff, s, q) in [(64, 15, 2), (60, 7, 3), (32, 9, 4)]:
        assert rel(oracle.w_kernel(0.1, 750.0, nff, s, q), P.w_kernel(0.1, 750.0, nff, s, q)) < 1e-12


# ---- properties -----------------------------------------------------------------------------------
def test_properties(oracle):
    rng = np.random.default_rng(17)
    N, W, Q, S, n = 40, 3, 4, 7, 400
    gcf, u, v, wb, vis = rand_case(rng, N, W, Q, S, n)
    z = lambda: np.zeros((N, N), dtype=np.complex128)
    G = oracle.convgrid2(gcf, z(), u, v, wb, vis)
    # linearity in vis
    vis2 = rng.normal(size=n) + 1j * rng.normal(size=n)
    Gs = oracle.convgrid2(gcf, z(), u, v, wb, 2.5 * vis + (0.5 - 1j) * vis2)
    assert rel(Gs, 2.5 * G + (0.5 - 1j) * oracle.convgrid2(gcf, z(), u, v, wb, vis2)) < 1e-13
    # permutation invariance
    perm = rng.permutation(n)
    assert rel(oracle.convgrid2(gcf, z(), u[perm], v[perm], wb[perm], vis[perm]), G) < 1e-13
    # accumulate-into semantics: permute (+) a
    G0 = rng.normal(size=(N, N)) + 0j
    assert rel(oracle.convgrid2(gcf, G0.copy(), u, v, wb, vis), G0 + G) < 1e-13
    # convgrid2 with W=1 == convgrid
    assert np.array_equal(oracle.convgrid2(gcf[:1], z(), u, v, np.zeros(n, np.int64), vis),
                          oracle.convgrid(gcf[0], z(), u, v, vis))
    # convgrid with Q=1,S=1 and unit kernel == grid away from rounding ties
    one = np.ones((1, 1, 1, 1), dtype=np.complex128)
    assert rel(oracle.convgrid(one, z(), u, v, vis), oracle.grid(z(), u, v, vis)) < 1e-15
    # out-of-range taps are dropped, never wrapped: far-outside points contribute nothing
    far_u = np.array([0.9, -0.9, 0.0, 3.0])
    far_v = np.array([0.0, 0.9, -0.9, -3.0])
    assert not oracle.convgrid2(gcf, z(), far_u, far_v, np.zeros(4, np.int64), np.ones(4, complex)).any()
    # a point whose footprint straddles the edge only fills the in-range part
    e = oracle.convgrid2(gcf, z(), np.array([-0.5]), np.array([0.0]), np.array([0]), np.array([1 + 0j]))
    assert e[:, S // 2 + 1:].any() == False and e[:, :S // 2 + 1].any()
    # adjoint identity <g, grid(vis)> == <degrid_{conj K}(g), vis>
    g = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    lhs = np.vdot(g, G)
    rhs = np.vdot(oracle.degrid2(np.conj(gcf), g, u, v, wb), vis)
    assert abs(lhs - rhs) / abs(lhs) < 1e-12


def test_w_cache_imaging_and_do_imaging_shapes():
    rng = np.random.default_rng(19)
    n = 60
    theta, lam = 0.05, 1280  # N = 64
    u, v = rng.uniform(-500, 500, n), rng.uniform(-500, 500, n)
    w = rng.uniform(-300, 300, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    fn = lambda th, la, uu, vv, ww, vs: P.w_cache_imaging(th, la, uu, vv, ww, vs, 100, 2, 32, 7)[0]
    img, psf, pmax = P.do_imaging(theta, lam, u, v, w, vis, fn)
    assert img.shape == (64, 64) and psf.shape == (64, 64)
    assert abs(psf.max() - 1.0) < 1e-12 and pmax > 0
