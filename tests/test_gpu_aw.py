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
                                         (72, 2, 2, 16, 4, 150), (90, 2, 2, 19, 3, 60), (64, 2, 2, 4, 5, 300)])
def test_awgrid_matches_oracle(ctx, oracle, N, W, Q, S, A, n):
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
