"""Dataset driver: the host-side mirror of the reference's src/ImageDataset.hs over libgridhip.so
and libgridhip_io.so — read visibilities and kernels from HDF5 in the reference's schema, image
them with the AW gridder on the GPU, write `/img`.

Schema (src/ImageDataset.hs:88-104, 112-146):
  /vis/vis (complex, any rank, flattened)   /vis/uvw (n,3)   /vis/antenna1, /vis/antenna2 (int64)
  /vis/time, /vis/frequency (double)
  /wkern/<theta %f>/<w>/kern   [Q,Q,S,S] per w-plane     /akern/<theta %f>/<ant>/<t>/<f>/kern   [S,S]

The real data files of the reference are Git-LFS stubs; `write_synthetic_dataset` writes files of
the same shape for tests and smoke runs.
"""
import numpy as np

from . import h5io

C_LIGHT = 299792458.0


def _fmt_theta(theta):
    return "%f" % theta  # printf "/wkern/%f" theta, src/ImageDataset.hs:112,139


def findClosestList(ws, w):
    """src/ImageDataset.hs:150-168 -> (value, index); hi starts at len-1 (the host twin of findClosest)."""
    lo, hi = 0, len(ws) - 1
    while (hi - lo) // 2 >= 1:
        mid = (hi + lo) // 2
        if w > ws[mid]:
            lo = mid
        else:
            hi = mid
    return (ws[lo], lo) if abs(w - ws[lo]) < abs(w - ws[hi]) else (ws[hi], hi)


def convertAndSort(names, conv=float):
    """src/ImageDataset.hs:174-178: parse the member names and sort numerically."""
    return sorted(((conv(s), s) for s in names), key=lambda t: t[0])


def uvw_lambda(f, uvw):
    """src/ImageDataset.hs:181-187"""
    return uvw * (f / C_LIGHT)


def getWKernels(file, theta):
    """src/ImageDataset.hs:136-148 -> (wkerns [W,Q,Q,S,S], wbins [W])"""
    base = "/wkern/" + _fmt_theta(theta)
    planes = convertAndSort(h5io.listGroupMembers(file, base))
    wk = h5io.readDatasets(file, [f"{base}/{s}/kern" for _, s in planes], np.complex128)
    return wk, np.array([w for w, _ in planes], dtype=np.float64)


def getAKernels(file, theta, t, f):
    """src/ImageDataset.hs:108-134 -> akerns [A,S,S]: for every antenna the kernel at the time and
    frequency closest to (t, f) among those stored for the first antenna.
    (The reference searches the frequency list with the *time* values, :125 `fs0 = map fst tsSorted`;
    with one time and one frequency per antenna, as in its data, both pick index 0 — here the
    frequency list is searched, which is what the surrounding code intends.)"""
    base = "/akern/" + _fmt_theta(theta)
    ants = convertAndSort(h5io.listGroupMembers(file, base), int)
    a0 = ants[0][1]
    ts = convertAndSort(h5io.listGroupMembers(file, f"{base}/{a0}"))
    tname = ts[findClosestList([x for x, _ in ts], t)[1]][1]
    fs = convertAndSort(h5io.listGroupMembers(file, f"{base}/{a0}/{tname}"))
    fname = fs[findClosestList([x for x, _ in fs], f)[1]][1]
    return h5io.readDatasets(file, [f"{base}/{a}/{tname}/{fname}/kern" for _, a in ants], np.complex128)


def readVis(file):
    return h5io.readDataset(file, "/vis/vis", np.complex128).reshape(-1)


def readBaselines(file):
    return h5io.readDataset(file, "/vis/uvw", np.float64)


def readSource(file):
    a1 = h5io.readDataset(file, "/vis/antenna1", np.int64)
    a2 = h5io.readDataset(file, "/vis/antenna2", np.int64)
    t = h5io.readDataset(file, "/vis/time", np.float64)
    f = h5io.readDataset(file, "/vis/frequency", np.float64)
    return a1, a2, t, float(f.reshape(-1)[0])


def aw_gridding(ctx, wfile, afile, datfile, n=None, outfile=None, theta=0.008, lam=300000):
    """src/ImageDataset.hs:29-86: read -> uvw_lambda -> doweight (on the UN-mirrored uvw, :59) ->
    mirror_uvw (:60) -> aw_imaging on vis*wt (:72-73) -> make_grid_hermitian -> real . ifft -> /img.
    Returns (image, max pixel).  theta / lam default to the reference's hard-coded values (:32-33)."""
    vis = readVis(datfile)
    uvw = readBaselines(datfile)
    a1, a2, ts, f = readSource(datfile)
    akerns = getAKernels(afile, theta, float(ts.reshape(-1)[0]), f)
    wkerns, wbins = getWKernels(wfile, theta)
    n = len(vis) if n is None else min(int(n), len(vis))
    uvw0 = uvw_lambda(f, uvw[:n])
    vis0 = vis[:n]
    cols = (uvw0[:, 0].copy(), uvw0[:, 1].copy(), uvw0[:, 2].copy())
    wt = ctx.doweight(theta, lam, cols, np.ones(n, dtype=np.complex128))
    uvw1, vis1 = ctx.mirror_uvw(cols, vis0)
    uvgrid = ctx.aw_imaging(theta, lam, wkerns, wbins, akerns, uvw1, (a1[:n], a2[:n], ts, f), vis1 * wt)
    img = np.real(ctx.ifft(ctx.make_grid_hermitian(uvgrid)))
    if outfile is not None:
        h5io.createh5File(outfile)
        h5io.createDataset(outfile, "/img", img)
    return img, float(img.max())


def write_synthetic_dataset(prefix, n=400, nant=4, nw=5, Q=2, S=15, theta=0.008, lam=300000, seed=0):
    """Small files in the reference's schema: <prefix>_vis.h5, <prefix>_wkern.h5, <prefix>_akern.h5."""
    rng = np.random.default_rng(seed)
    f = 1.0e8
    N = int(np.rint(theta * lam))
    span = 0.35 * N / theta * (C_LIGHT / f)  # metres, so that uvw_lambda lands inside the grid
    uvw = rng.uniform(-span, span, (n, 3))
    uvw[:, 2] = rng.uniform(-200.0, 200.0, n) * (C_LIGHT / f)
    vis = (rng.normal(size=(n, 1, 1)) + 1j * rng.normal(size=(n, 1, 1))).astype(np.complex128)
    a1 = rng.integers(0, nant, n).astype(np.int64)
    a2 = rng.integers(0, nant, n).astype(np.int64)
    visf, wf, af = prefix + "_vis", prefix + "_wkern", prefix + "_akern"
    h5io.createh5File(visf)
    h5io.createDataset(visf, "/vis/vis", vis)
    h5io.createDataset(visf, "/vis/uvw", uvw)
    h5io.createDataset(visf, "/vis/antenna1", a1)
    h5io.createDataset(visf, "/vis/antenna2", a2)
    h5io.createDataset(visf, "/vis/time", np.full(n, 58000.25))
    h5io.createDataset(visf, "/vis/frequency", np.full(n, f))
    h5io.createh5File(wf)
    jj = np.arange(S) - S // 2
    r2 = (jj[:, None] ** 2 + jj[None, :] ** 2).astype(np.float64)
    wvals = np.linspace(-200.0, 200.0, nw)
    for w in wvals:
        k = np.empty((Q, Q, S, S), dtype=np.complex128)
        for yf in range(Q):
            for xf in range(Q):
                k[yf, xf] = np.exp(-r2 / 20.0) * np.exp(1j * 1e-3 * w * (r2 + yf + 0.5 * xf)) / 50.0
        h5io.createDataset(wf, f"/wkern/{_fmt_theta(theta)}/{float(w)!r}/kern", k)
    h5io.createh5File(af)
    for a in range(nant):
        k = (np.exp(-r2 / (10.0 + a)) * (1.0 + 0.1j * a) / 30.0).astype(np.complex128)
        h5io.createDataset(af, f"/akern/{_fmt_theta(theta)}/{a}/58000.0/{f!r}/kern", k)
    return visf, wf, af
