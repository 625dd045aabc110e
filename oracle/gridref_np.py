"""numpy restatement of src/Gridding.hs — CPU ORACLE #2 (test infrastructure, NOT product code).

An implementation independent of oracle/gridref.c (vectorised, numpy FFTs) used only to
cross-check the C oracle and to generate the small golden fixtures under tests/golden/.
Only tests/ and oracle/make_golden.py import it.  Parity status: see oracle/gridref.h —
one recorded KAT from the reference (old/BrokenNumbers.hs:86-91); everything else is
"parity unpinned" by the reference and pinned by the two restatements agreeing.

All citations are to /root/reference/src/Gridding.hs unless stated otherwise.
"""
import numpy as np


def haskell_round(x):
    """Prelude `round` = round-half-even (used on the HOST for N = round(theta*lam), :87,:118,:416)."""
    return int(np.rint(x))


def acc_round(x):
    """Accelerate `round` as lowered by the LLVM backends = libm round (half away from zero)."""
    x = np.asarray(x, dtype=np.float64)
    return (np.sign(x) * np.floor(np.abs(x) + 0.5)).astype(np.int64)


def frac_coord(n, qpx, p):
    """:126-140"""
    p = np.asarray(p, dtype=np.float64)
    halfnf = np.float64(n // 2)
    nf = np.float64(n)
    qpxf = np.float64(qpx)
    x = halfnf + p * nf
    flx = np.floor(x + 0.5 / qpxf).astype(np.int64)
    fracx = acc_round((x - flx.astype(np.float64)) * qpxf)
    return flx, np.clip(fracx, 0, qpx - 1)


def frac_coords(hw, qpx, u, v):
    """:142-151 — returns (x, xf, y, yf)"""
    h, w = hw
    x, xf = frac_coord(w, qpx, u)
    y, yf = frac_coord(h, qpx, v)
    return x, xf, y, yf


def grid(G, u, v, vis):
    """:95-112 (cells out of range dropped)"""
    H, Wd = G.shape
    halfn = H // 2
    x = halfn + np.floor(0.5 + np.float64(H) * np.asarray(u)).astype(np.int64)
    y = halfn + np.floor(0.5 + np.float64(H) * np.asarray(v)).astype(np.int64)
    ok = (x >= 0) & (y >= 0) & (x < Wd) & (y < H)
    np.add.at(G, (y[ok], x[ok]), np.asarray(vis)[ok])
    return G


def convgrid2(gcf, G, u, v, wbin, vis):
    """:199-244 ; gcf [W,Q,Q,gh,gw]"""
    W, Q, _, gh, gw = gcf.shape
    H, Wd = G.shape
    x, xf, y, yf = frac_coords((H, Wd), Q, u, v)
    x0 = x - gw // 2
    y0 = y - gh // 2
    vis = np.asarray(vis, dtype=np.complex128)
    wbin = np.asarray(wbin, dtype=np.int64)
    for i in range(gh):
        for j in range(gw):
            xx = x0 + j
            yy = y0 + i
            ok = (xx >= 0) & (yy >= 0) & (xx < Wd) & (yy < H)  # fixoutofbounds :883-891
            val = vis * gcf[wbin, yf, xf, i, j]
            np.add.at(G, (yy[ok], xx[ok]), val[ok])
    return G


def convgrid(gcf, G, u, v, vis):
    """:153-197 ; gcf [Q,Q,gh,gw]"""
    return convgrid2(gcf[None], G, u, v, np.zeros(len(np.atleast_1d(u)), dtype=np.int64), vis)


def degrid2(gcf, G, u, v, wbin):
    """adjoint-pattern gather (not in the reference)."""
    W, Q, _, gh, gw = gcf.shape
    H, Wd = G.shape
    x, xf, y, yf = frac_coords((H, Wd), Q, u, v)
    x0 = x - gw // 2
    y0 = y - gh // 2
    out = np.zeros(len(x), dtype=np.complex128)
    wbin = np.asarray(wbin, dtype=np.int64)
    for i in range(gh):
        for j in range(gw):
            xx = x0 + j
            yy = y0 + i
            ok = (xx >= 0) & (yy >= 0) & (xx < Wd) & (yy < H)
            g = np.zeros(len(x), dtype=np.complex128)
            g[ok] = G[yy[ok], xx[ok]]
            out += gcf[wbin, yf, xf, i, j] * g
    return out


def find_closest(ws, w):
    """:895-907 (hi clamped to len-1)"""
    ws = np.asarray(ws, dtype=np.float64)
    lo, hi = 0, len(ws)
    while (hi - lo) // 2 >= 1:
        mid = (hi + lo) // 2
        if w > ws[mid]:
            lo = mid
        else:
            hi = mid
    hc = min(hi, len(ws) - 1)
    return lo if abs(w - ws[lo]) < abs(w - ws[hc]) else hc


def wbins(w, wstep):
    """:426-432 — returns (wbin, wmin, nplanes)"""
    rw = wstep * acc_round(np.asarray(w, dtype=np.float64) / np.float64(wstep))
    mn, mx = int(rw.min()), int(rw.max())
    return (rw - mn) // wstep, mn, (mx - mn) // wstep + 1


def mirror_uvw(u, v, w, vis):
    """:551-562"""
    neg = np.asarray(v) < 0
    s = np.where(neg, -1.0, 1.0)
    return u * s, v * s, w * s, np.where(neg, np.conj(vis), vis)


def doweight(N, pu, pv, vis):
    """:564-583 (p already divided by lam)"""
    x, _, y, _ = frac_coords((N, N), 1, pu, pv)
    ok = (x >= 0) & (y >= 0) & (x < N) & (y < N)
    cnt = np.zeros((N, N), dtype=np.float64)
    np.add.at(cnt, (y[ok], x[ok]), 1.0)
    out = np.array(vis, dtype=np.complex128)
    out[ok] = out[ok] / cnt[y[ok], x[ok]]
    return out


def make_grid_hermitian(G):
    """:585-605"""
    N = G.shape[0]
    if N % 2 == 0:
        add = np.zeros_like(G)
        add[1:, 1:] = np.conj(G[1:, 1:][::-1, ::-1])  # G[N-y, N-x], x,y != 0
    else:
        add = np.conj(G[::-1, ::-1])
    return G + add


def shift2d(a):
    """accelerate-fft DFT.Centre shift2D == numpy fftshift [upstream, un-vendored]"""
    return np.fft.fftshift(a)


def ishift2d(a):
    return np.fft.ifftshift(a)


def ifft_c(a):
    """ifft, :828-829 — accelerate-fft Inverse is 1/N normalised like numpy's ifft2"""
    return shift2d(np.fft.ifft2(ishift2d(a)))


def fft_c(a):
    return shift2d(np.fft.fft2(ishift2d(a)))


def pad_mid(ff, n):
    """:682-691 with padder :863-877 — padder transposes the input (index2 oldx oldy, :875)"""
    n0 = ff.shape[0]
    if n == n0:
        return ff
    p0 = n // 2 - n0 // 2
    out = np.zeros((n, n), dtype=ff.dtype)
    out[p0:p0 + n0, p0:p0 + n0] = ff.T
    return out


def extract_mid(a, n):
    """:694-707"""
    c = a.shape[0] // 2
    s = n // 2
    return a[c - s:c - s + n, c - s:c - s + n]


def convolve2d(a1, a2):
    """:795-811"""
    n = a1.shape[0]
    m = 1
    while m < 2 * n - 1:
        m *= 2
    f1 = np.fft.ifft2(ishift2d(pad_mid(a1, m)))
    f2 = np.fft.ifft2(ishift2d(pad_mid(a2, m)))
    conv = shift2d(np.fft.fft2(f1 * f2))
    return extract_mid(conv, n) * np.float64(m * m)


def same_conv_direct(a, b):
    """centred 'same' linear convolution (reference-free definition, for the quirk test)"""
    n = a.shape[0]
    c = n // 2
    out = np.zeros((n, n), dtype=np.complex128)
    for i in range(n):
        for j in range(n):
            # a[i,j] * b[y-i+c, x-j+c]
            ylo, yhi = max(0, i - c), min(n, n + i - c)
            xlo, xhi = max(0, j - c), min(n, n + j - c)
            out[ylo:yhi, xlo:xhi] += a[i, j] * b[ylo - i + c:yhi - i + c, xlo - j + c:xhi - j + c]
    return out


def aw_kernel_fn2(yf, xf, wkern, a1, a2):
    """:761-775 ; wkern [Q,Q,S,S]"""
    return convolve2d(convolve2d(a1, a2), wkern[yf, xf])


def awgrid(wkerns, akerns, G, u, v, wbin, a1, a2, vis):
    """convgrid3 / convgrid4, :246-396"""
    W, Q, _, S, _ = wkerns.shape
    H, Wd = G.shape
    x, xf, y, yf = frac_coords((H, Wd), Q, u, v)
    for k in range(len(x)):
        aw = np.conj(aw_kernel_fn2(yf[k], xf[k], wkerns[wbin[k]], akerns[a1[k]], akerns[a2[k]]))
        for i in range(S):
            for j in range(S):
                xx = x[k] - S // 2 + j
                yy = y[k] - S // 2 + i
                if 0 <= xx < Wd and 0 <= yy < H:
                    G[yy, xx] += vis[k] * aw[i, j]
    return G


def w_kernel(theta, w, npixFF, npixKern, qpx):
    """:610-728 — returns [Q,Q,S,S]"""
    n = npixFF
    step = 1.0 / n
    base = (-(n // 2)) * step + np.arange(n, dtype=np.float64) * step
    l = base[None, :] * theta
    m = base[:, None] * theta
    r2 = l * l + m * m
    ph = 1.0 - np.sqrt(1.0 - r2)
    ff = np.exp(1j * (2.0 * np.pi * w * ph))
    af = ifft_c(pad_mid(ff, n * qpx))
    na = af.shape[0]
    s = npixKern
    c = na // 2 - qpx * (s // 2)
    out = np.empty((qpx, qpx, s, s), dtype=np.complex128)
    for yf in range(qpx):
        for xf in range(qpx):
            out[yf, xf] = af[c - yf:c - yf + qpx * s:qpx, c - xf:c - xf + qpx * s:qpx]
    return out * np.float64(qpx * qpx)


def w_cache_imaging(theta, lam, u, v, w, vis, wstep, qpx, npixFF, npixKern):
    """:399-449 — returns (grid, kernels, wbin)"""
    N = haskell_round(theta * lam)
    wb, wmin, steps = wbins(w, wstep)
    kerns = np.stack([np.conj(w_kernel(theta, float(i * wstep + wmin), npixFF, npixKern, qpx))
                      for i in range(steps)])
    G = np.zeros((N, N), dtype=np.complex128)
    convgrid2(kerns, G, np.asarray(u) / np.float64(lam), np.asarray(v) / np.float64(lam), wb, vis)
    return G, kerns, wb


def do_imaging(theta, lam, u, v, w, vis, imgfn):
    """:509-549 — imgfn(theta, lam, u, v, w, vis) -> complex grid. Returns (image, psf, pmax)."""
    u1, v1, w1, vis1 = mirror_uvw(np.asarray(u), np.asarray(v), np.asarray(w), np.asarray(vis))
    N = haskell_round(theta * lam)
    wt = doweight(N, u1 / np.float64(lam), v1 / np.float64(lam), np.ones(len(u1), dtype=np.complex128))
    cdrt = imgfn(theta, lam, u1, v1, w1, wt * vis1)
    drt = np.real(ifft_c(make_grid_hermitian(cdrt)))
    c = imgfn(theta, lam, u1, v1, w1, wt)
    psf = np.real(ifft_c(make_grid_hermitian(c)))
    pmax = psf.max()
    return drt / pmax, psf / pmax, pmax
