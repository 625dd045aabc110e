"""ctypes loader for oracle/libgridref.so — CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64 = C.c_int64
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int64)


def build():
    """Compile the C restatement (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libgridref.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgridref.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.gridref_find_closest.restype = i64
        _LIB.gridref_max_threads.restype = C.c_int
    return _LIB


def _d(a):
    return a.ctypes.data_as(dp)


def _i(a):
    return a.ctypes.data_as(ip) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def frac_coord(n, qpx, p):
    p = _f64(p)
    flx = np.empty(len(p), dtype=np.int64)
    fr = np.empty(len(p), dtype=np.int64)
    lib().gridref_frac_coord(i64(n), i64(qpx), i64(len(p)), _d(p), _i(flx), _i(fr))
    return flx, fr


def grid(G, u, v, vis):
    assert G.dtype == np.complex128 and G.flags.c_contiguous
    u, v, vis = _f64(u), _f64(v), _c128(vis)
    lib().gridref_grid(i64(G.shape[0]), i64(G.shape[1]), _d(G), i64(len(u)), _d(u), _d(v), _d(vis))
    return G


def convgrid2(gcf, G, u, v, wbin, vis, mt_mode=None, nthreads=0):
    assert G.dtype == np.complex128 and G.flags.c_contiguous
    gcf = _c128(gcf)
    W, Q, Q2, gh, gw = gcf.shape
    assert Q == Q2
    u, v, vis, wbin = _f64(u), _f64(v), _c128(vis), _i64(wbin)
    args = [i64(G.shape[0]), i64(G.shape[1]), _d(G), i64(len(u)), i64(W), i64(Q), i64(gh), i64(gw),
            _d(gcf), _d(u), _d(v), _i(wbin), _d(vis)]
    if mt_mode is None:
        lib().gridref_convgrid2(*args)
    else:
        lib().gridref_convgrid2_mt(*args, C.c_int(mt_mode), C.c_int(nthreads))
    return G


def convgrid(gcf, G, u, v, vis):
    gcf = _c128(gcf)
    Q, _, gh, gw = gcf.shape
    u, v, vis = _f64(u), _f64(v), _c128(vis)
    lib().gridref_convgrid(i64(G.shape[0]), i64(G.shape[1]), _d(G), i64(len(u)), i64(Q), i64(gh),
                           i64(gw), _d(gcf), _d(u), _d(v), _d(vis))
    return G


def degrid2(gcf, G, u, v, wbin):
    gcf, G = _c128(gcf), _c128(G)
    W, Q, _, gh, gw = gcf.shape
    u, v, wbin = _f64(u), _f64(v), _i64(wbin)
    out = np.empty(len(u), dtype=np.complex128)
    lib().gridref_degrid2(i64(G.shape[0]), i64(G.shape[1]), _d(G), i64(len(u)), i64(W), i64(Q),
                          i64(gh), i64(gw), _d(gcf), _d(u), _d(v), _i(wbin), _d(out))
    return out


def find_closest(ws, w):
    ws = _f64(ws)
    return int(lib().gridref_find_closest(i64(len(ws)), _d(ws), C.c_double(w)))


def wbins(w, wstep):
    w = _f64(w)
    out = np.empty(len(w), dtype=np.int64)
    mn, npl = i64(0), i64(0)
    lib().gridref_wbins(i64(len(w)), _d(w), i64(wstep), _i(out), C.byref(mn), C.byref(npl))
    return out, mn.value, npl.value


def mirror_uvw(u, v, w, vis):
    u, v, w, vis = _f64(u).copy(), _f64(v).copy(), _f64(w).copy(), _c128(vis).copy()
    lib().gridref_mirror_uvw(i64(len(u)), _d(u), _d(v), _d(w), _d(vis))
    return u, v, w, vis


def doweight(N, pu, pv, vis):
    pu, pv, vis = _f64(pu), _f64(pv), _c128(vis).copy()
    lib().gridref_doweight(i64(N), i64(len(pu)), _d(pu), _d(pv), _d(vis))
    return vis


def make_grid_hermitian(G):
    G = _c128(G).copy()
    lib().gridref_make_grid_hermitian(i64(G.shape[0]), _d(G))
    return G


def convolve2d(a1, a2, direct=False):
    a1, a2 = _c128(a1), _c128(a2)
    out = np.empty_like(a1)
    fn = lib().gridref_convolve2d_direct if direct else lib().gridref_convolve2d
    fn(i64(a1.shape[0]), _d(a1), _d(a2), _d(out))
    return out


def aw_kernel_fn2(yf, xf, wkern, a1, a2, direct=False):
    wkern, a1, a2 = _c128(wkern), _c128(a1), _c128(a2)
    Q, _, S, _ = wkern.shape
    out = np.empty((S, S), dtype=np.complex128)
    lib().gridref_aw_kernel_fn2(i64(Q), i64(S), i64(yf), i64(xf), _d(wkern), _d(a1), _d(a2), _d(out),
                                C.c_int(int(direct)))
    return out


def awgrid(wkerns, akerns, G, u, v, wbin, a1, a2, vis, direct=False):
    wkerns, akerns = _c128(wkerns), _c128(akerns)
    W, Q, _, S, _ = wkerns.shape
    u, v, vis = _f64(u), _f64(v), _c128(vis)
    wbin, a1, a2 = _i64(wbin), _i64(a1), _i64(a2)
    lib().gridref_awgrid(i64(G.shape[0]), i64(G.shape[1]), _d(G), i64(len(u)), i64(W), i64(Q), i64(S),
                         i64(akerns.shape[0]), _d(wkerns), _d(akerns), _d(u), _d(v), _i(wbin), _i(a1),
                         _i(a2), _d(vis), C.c_int(int(direct)))
    return G


def w_kernel(theta, w, npixFF, npixKern, qpx):
    out = np.empty((qpx, qpx, npixKern, npixKern), dtype=np.complex128)
    rc = lib().gridref_w_kernel(C.c_double(theta), C.c_double(w), i64(npixFF), i64(npixKern), i64(qpx),
                                _d(out))
    assert rc == 0
    return out


def fft2_centered(a, inverse):
    a = _c128(a)
    out = np.empty_like(a)
    rc = lib().gridref_fft2_centered(i64(a.shape[0]), _d(a), _d(out), C.c_int(int(inverse)))
    assert rc == 0
    return out


def max_threads():
    return int(lib().gridref_max_threads())
