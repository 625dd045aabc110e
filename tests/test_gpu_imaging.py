"""GPU parity for the callers either side of the gridder: the w-bin rule,
findClosest, mirror_uvw, doweight, make_grid_hermitian, the centred (i)FFT (hipFFT), the w-kernel
generator, the imaging functions and do_imaging — each against the CPU oracle.

None of these is pinned by a reference-recorded output ("parity unpinned"); the
parameter sets follow the reference's test scripts (theta=0.1, Q=2, S=31, test/GridTesting.hs:85-93)
scaled down so the oracle runs in seconds.  Tolerance 1e-10 relative unless stated (FFT-based
quantities accumulate ~log2(N) roundings: observed ~1e-15)."""
import numpy as np
import pytest

from oracle import gridref_np as P

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_wbins_and_find_closest(ctx, oracle):
    rng = np.random.default_rng(41)
    w = rng.uniform(-9000, 21000, 100000)
    wb, mn, npl = ctx.wbins(w, 2000)
    rb, rmn, rnpl = oracle.wbins(w, 2000)
    assert np.array_equal(wb, rb) and (mn, npl) == (rmn, rnpl)
    ws = np.sort(rng.uniform(0, 100, 129))
    x = rng.uniform(-10, 120, 5000)
    got = ctx.findClosest(ws, x)
    ref = np.array([oracle.find_closest(ws, xi) for xi in x])
    assert np.array_equal(got, ref)


def test_mirror_doweight_hermitian(ctx, oracle):
    rng = np.random.default_rng(42)
    n = 20000
    u, v, w = rng.uniform(-500, 500, n), rng.uniform(-500, 500, n), rng.uniform(-300, 300, n)
    vis = rng.normal(size=n) + 1j * rng.normal(size=n)
    (mu, mv, mw), mvis = ctx.mirror_uvw((u, v, w), vis)
    ru, rv, rw, rvis = oracle.mirror_uvw(u, v, w, vis)
    assert all(np.array_equal(a, b) for a, b in ((mu, ru), (mv, rv), (mw, rw), (mvis, rvis)))
    theta, lam = 0.05, 2560  # N = 128
    N = ctx.image_size(theta, lam)
    assert N == 128 == P.haskell_round(theta * lam)
    got = ctx.doweight(theta, lam, (u, v, w), vis)
    ref = oracle.doweight(N, u / lam, v / lam, vis)
    assert rel(got, ref) < 1e-15
    for n_ in (64, 65):
        g = rng.normal(size=(n_, n_)) + 1j * rng.normal(size=(n_, n_))
        assert np.array_equal(ctx.make_grid_hermitian(g), oracle.make_grid_hermitian(g))


@pytest.mark.parametrize("N", [64, 100, 127, 240])
def test_centred_fft(ctx, N):
    rng = np.random.default_rng(N)
    a = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    assert rel(ctx.ifft(a), P.ifft_c(a)) < TOL
    assert rel(ctx.fft(a), P.fft_c(a)) < TOL


@pytest.mark.parametrize("npixFF,S,Q,w", [(256, 31, 2, 100.0), (256, 31, 2, 1000.0), (64, 15, 8, 750.0), (60, 7, 3, 5000.0)])
def test_w_kernel(ctx, golden, npixFF, S, Q, w):
    got = ctx.w_kernel(0.1, w, npixFF, S, Q)
    assert rel(got, P.w_kernel(0.1, w, npixFF, S, Q)) < TOL
    if (npixFF, S, Q) == (256, 31, 2):
        assert rel(got, golden(f"wkernel_w{int(w)}")["expected"]) < TOL


def _vis(seed, n, span, wspan):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-span, span, n), rng.uniform(-span, span, n), rng.uniform(-wspan, wspan, n),
            rng.normal(size=n) + 1j * rng.normal(size=n))


def test_simple_and_conv_imaging(ctx, oracle):
    theta, lam = 0.05, 2560
    N = 128
    u, v, w, vis = _vis(5, 5000, 1300, 100)  # some fall outside the grid
    got = ctx.simple_imaging(theta, lam, (u, v, w), None, vis)
    ref = oracle.grid(np.zeros((N, N), dtype=np.complex128), u / lam, v / lam, vis)
    assert rel(got, ref) < TOL
    rng = np.random.default_rng(6)
    kv = rng.normal(size=(4, 4, 9, 9)) + 1j * rng.normal(size=(4, 4, 9, 9))
    got = ctx.conv_imaging(kv, theta, lam, np.stack([u, v, w], 1), None, vis)  # (n,3) matrix layout
    ref = oracle.convgrid(kv, np.zeros((N, N), dtype=np.complex128), u / lam, v / lam, vis)
    assert rel(got, ref) < TOL


def test_w_cache_imaging(ctx):
    theta, lam = 0.05, 2560
    u, v, w, vis = _vis(7, 3000, 1200, 900)
    ko = dict(wstep=100, qpx=2, npixFF=64, npixKern=15)
    got = ctx.w_cache_imaging(ko, theta, lam, (u, v, w), None, vis)
    ref, kerns, wb = P.w_cache_imaging(theta, lam, u, v, w, vis, 100, 2, 64, 15)
    assert kerns.shape[0] == wb.max() + 1 > 10
    assert rel(got, ref) < TOL


@pytest.mark.parametrize("kind", ["simple", "conv", "w_cache"])
def test_do_imaging(ctx, kind):
    theta, lam = 0.05, 2560
    u, v, w, vis = _vis(9, 4000, 1100, 500)
    rng = np.random.default_rng(10)
    kv = rng.normal(size=(2, 2, 7, 7)) + 1j * rng.normal(size=(2, 2, 7, 7))
    ko = dict(wstep=200, qpx=2, npixFF=32, npixKern=7)
    N = 128
    if kind == "simple":
        fn = lambda th, la, uu, vv, ww, vs: P.grid(np.zeros((N, N), complex), uu / la, vv / la, vs)
        spec = ("simple",)
    elif kind == "conv":
        fn = lambda th, la, uu, vv, ww, vs: P.convgrid(kv, np.zeros((N, N), complex), uu / la, vv / la, vs)
        spec = ("conv", kv)
    else:
        fn = lambda th, la, uu, vv, ww, vs: P.w_cache_imaging(th, la, uu, vv, ww, vs, 200, 2, 32, 7)[0]
        spec = ("w_cache", ko)
    rimg, rpsf, rpmax = P.do_imaging(theta, lam, u, v, w, vis, fn)
    img, psf, pmax = ctx.do_imaging(theta, lam, (u, v, w), None, None, None, 1.0e8, vis, spec)
    assert abs(pmax - rpmax) / abs(rpmax) < TOL
    assert rel(psf, rpsf) < TOL and abs(psf.max() - 1.0) < 1e-12
    assert rel(img, rimg) < TOL


@pytest.mark.parametrize("kind", ["simple", "w_cache"])
def test_do_imaging_at_the_drivers_size_host_and_resident(ctx, kind):
    """do_imaging (src/Gridding.hs:509-549) at the size the reference's driver images at - theta = 0.008, lam = 300000,
    N = round(theta * lam) = 2400 (src/ImageDataset.hs:32-33; not a power of two: hipFFT's general path) - with
    1.2 x 10^5 visibilities and, for w_cache_imaging, 15 x 15 kernels generated on the device, against the numpy oracle;
    and the device-resident form (gridhip_do_imaging_dev: torch cuda tensors in and out, (n, 3) uvw matrix) against
    the host form.  Parity unpinned by the reference (it records no image)."""
    import torch
    theta, lam = 0.008, 300000
    N = ctx.image_size(theta, lam)
    assert N == 2400
    n = 120_000
    u, v, w, vis = _vis(31, n, 0.47 * lam, 700.0)
    ko = dict(wstep=100, qpx=2, npixFF=64, npixKern=15)
    if kind == "simple":
        fn = lambda th, la, uu, vv, ww, vs: P.grid(np.zeros((N, N), complex), uu / la, vv / la, vs)
        spec = ("simple",)
    else:
        fn = lambda th, la, uu, vv, ww, vs: P.w_cache_imaging(th, la, uu, vv, ww, vs, 100, 2, 64, 15)[0]
        spec = ("w_cache", ko)
    rimg, rpsf, rpmax = P.do_imaging(theta, lam, u, v, w, vis, fn)
    img, psf, pmax = ctx.do_imaging(theta, lam, (u, v, w), None, None, None, 1.0e8, vis, spec)
    assert abs(pmax - rpmax) / abs(rpmax) < TOL
    assert rel(psf, rpsf) < TOL and abs(psf.max() - 1.0) < 1e-12
    assert rel(img, rimg) < TOL
    dev = torch.device("cuda:0")
    uvw = torch.from_numpy(np.stack([u, v, w], 1)).to(dev)
    tvis = torch.from_numpy(vis).to(dev)
    for _ in range(2):  # (the second call draws every block from the context's pool)
        dimg, dpsf, dpmax = ctx.do_imaging(theta, lam, uvw, None, None, None, 1.0e8, tvis, spec)
        torch.cuda.synchronize()
        assert dimg.is_cuda and dpsf.is_cuda
        assert abs(dpmax - rpmax) / abs(rpmax) < TOL
        assert rel(dimg.cpu().numpy(), rimg) < TOL and rel(dpsf.cpu().numpy(), rpsf) < TOL
    assert torch.equal(tvis.cpu(), torch.from_numpy(vis))  # inputs are not modified (mirror works on a copy)
    assert ctx.get_option("errors") == 0


def test_w_cache_imaging_resident(ctx):
    import torch
    theta, lam = 0.05, 2560
    u, v, w, vis = _vis(7, 3000, 1200, 900)
    ko = dict(wstep=100, qpx=2, npixFF=64, npixKern=15)
    ref = ctx.w_cache_imaging(ko, theta, lam, (u, v, w), None, vis)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    got = ctx.w_cache_imaging(ko, theta, lam, (t(u), t(v), t(w)), None, t(vis))
    got3 = ctx.w_cache_imaging(ko, theta, lam, t(np.stack([u, v, w], 1)), None, t(vis))
    torch.cuda.synchronize()
    assert rel(got.cpu().numpy(), ref) < 1e-13 and rel(got3.cpu().numpy(), ref) < 1e-13


def test_w_kernel_table_kept_between_calls_is_never_stale(ctx):
    """The context keeps the last w-kernel table w_cache_imaging built and reuses it when the field of view, the planes
    and the kernel's shape match (gridhip_ctx::wk_cache).  Calls that alternate between geometries - another field of
    view, another w range (other planes), another support, another oversampling - must each come out exactly as a
    fresh context computes them, and a repeated geometry exactly as the first time."""
    import gridhip
    lam = 2560
    u, v, w, vis = _vis(11, 2500, 1200, 900)
    cases = [(0.05, dict(wstep=100, qpx=2, npixFF=64, npixKern=15), w),
             (0.04, dict(wstep=100, qpx=2, npixFF=64, npixKern=15), w),          # another field of view
             (0.05, dict(wstep=100, qpx=2, npixFF=64, npixKern=15), w * 0.5),    # other planes
             (0.05, dict(wstep=100, qpx=2, npixFF=64, npixKern=9), w),           # another support
             (0.05, dict(wstep=100, qpx=4, npixFF=64, npixKern=15), w),          # another oversampling
             (0.05, dict(wstep=50, qpx=2, npixFF=64, npixKern=15), w)]           # another plane spacing
    fresh = []
    for theta, ko, ww in cases:
        c = gridhip.Context(0)
        fresh.append(c.w_cache_imaging(ko, theta, lam, (u, v, ww), None, vis))
        c.close()
    order = [0, 0, 1, 0, 2, 2, 3, 4, 5, 0, 5, 1]
    for k in order:
        theta, ko, ww = cases[k]
        got = ctx.w_cache_imaging(ko, theta, lam, (u, v, ww), None, vis)
        assert np.array_equal(got, fresh[k]) or rel(got, fresh[k]) < 1e-13, k
    ref, _, _ = P.w_cache_imaging(cases[2][0], lam, u, v, cases[2][2], vis, 100, 2, 64, 15)
    assert rel(fresh[2], ref) < TOL
