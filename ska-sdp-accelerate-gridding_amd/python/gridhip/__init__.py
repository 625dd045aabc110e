"""gridhip — Python binding of libgridhip.so, the MI355X-native w-projection gridder.

The module mirrors the gridder interface of the reference's src/Gridding.hs — same names,
argument order and meaning:

    grid      a p v            (src/Gridding.hs:95-98)
    convgrid  gcf a p v        (src/Gridding.hs:153-157)
    convgrid2 gcf a p wbin v   (src/Gridding.hs:199-204)
    degrid2   gcf a p wbin     (the gather twin; absent from the reference)

`a` is the destination grid (complex128, [H, W], ACCUMULATED INTO and returned), `p` the
baselines already scaled to (-.5, .5) — a (u, v, w) tuple of float64 arrays or an (n, 3)
array — `v` the visibilities (complex128).  numpy arguments take the synchronous host path of
the C ABI; torch CUDA tensors take the asynchronous device path on torch's current stream.

There is no CPU fallback: importing works anywhere the library is built, computing needs a
gfx950 GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GridHipError, LIB_PATH  # noqa: F401

__all__ = ["Context", "default_context", "grid", "convgrid", "convgrid2", "degrid2", "GridHipError"]


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _split_p(p):
    """(u, v, w) tuple or (n, 3) array -> (u, v, element stride)."""
    if isinstance(p, (tuple, list)):
        return p[0], p[1], 1
    if p.ndim == 2 and p.shape[1] == 3:
        if _is_torch(p):
            import torch
            p = p.to(torch.float64).contiguous()  # (the library reads doubles at stride 3)
            return p[:, 0], p[:, 1], 3
        p = np.ascontiguousarray(p, dtype=np.float64)
        return p[:, 0], p[:, 1], 3
    raise ValueError("p must be a (u, v, w) tuple or an (n, 3) array")


class Context:
    """One device + one stream (gridhip_ctx).  Not thread-safe."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.gridhip_create(int(device), C.byref(h))
        if rc != 0:
            raise GridHipError(rc, self._lib.gridhip_strerror(rc).decode())
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gridhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise GridHipError(rc, self._lib.gridhip_last_error(self._h).decode() or
                               self._lib.gridhip_strerror(rc).decode())

    def set_option(self, key, value):
        self._check(self._lib.gridhip_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64()
        self._check(self._lib.gridhip_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def set_stream(self, stream_ptr):
        """Enqueue on this hipStream_t; 0/None is HIP's default (null) stream."""
        self._check(self._lib.gridhip_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def reset_stream(self):
        self._check(self._lib.gridhip_reset_stream(self._h))

    def synchronize(self):
        self._check(self._lib.gridhip_synchronize(self._h))

    def enable_timing(self, on=True):
        self._check(self._lib.gridhip_enable_timing(self._h, int(bool(on))))

    def last_timing(self):
        """(total_ms, prepass_ms, kernel_ms) of the last device call, from HIP events on the stream."""
        t, p, k = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.gridhip_last_timing(self._h, C.byref(t), C.byref(p), C.byref(k)))
        return t.value, p.value, k.value

    def timing(self, back=0):
        """(total_ms, prepass_ms, kernel_ms) of the timed device call `back` calls before the last one."""
        t, p, k = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.gridhip_timing(self._h, int(back), C.byref(t), C.byref(p), C.byref(k)))
        return t.value, p.value, k.value

    def last_dropped(self):
        d = C.c_int64()
        self._check(self._lib.gridhip_last_dropped(self._h, C.byref(d)))
        return d.value

    def _use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    # -- argument marshalling -----------------------------------------------------------------
    @staticmethod
    def _np(x, dt):
        return np.ascontiguousarray(x, dtype=dt)

    @staticmethod
    def _ptr(x):
        if x is None:
            return None
        if _is_torch(x):
            return C.c_void_p(x.data_ptr())
        return C.c_void_p(x.ctypes.data)

    def _prep(self, a, p, vis, wbin, gcf):
        """Normalise one call's arguments; returns (dev, a, u, v, stride, vis, wbin, gcf)."""
        dev = _is_torch(a)
        u, v, stride = _split_p(p)
        if dev:
            import torch
            assert a.is_cuda and a.dtype == torch.complex128 and a.is_contiguous(), "grid must be a contiguous cuda complex128 tensor"
            f = lambda t, dt: None if t is None else (t if (t.dtype == dt and (t.is_contiguous() or stride == 3)) else t.to(dt).contiguous())
            if stride == 1:
                u, v = u.to(torch.float64).contiguous(), v.to(torch.float64).contiguous()
            vis = f(vis, torch.complex128)
            wbin = f(wbin, torch.int64)
            gcf = f(gcf, torch.complex128)
            self._use_torch_stream()
        else:
            if not (isinstance(a, np.ndarray) and a.dtype == np.complex128 and a.flags.c_contiguous):
                raise ValueError("grid must be a C-contiguous complex128 ndarray (it is accumulated in place)")
            if stride == 1:
                u, v = self._np(u, np.float64), self._np(v, np.float64)
            vis = None if vis is None else self._np(vis, np.complex128)
            wbin = None if wbin is None else self._np(wbin, np.int64)
            gcf = None if gcf is None else self._np(gcf, np.complex128)
        return dev, a, u, v, stride, vis, wbin, gcf

    # -- the gridders ---------------------------------------------------------------------------
    def grid(self, a, p, v):
        """src/Gridding.hs:95-112"""
        dev, a, pu, pv, stride, vis, _, _ = self._prep(a, p, v, None, None)
        n = int(pu.shape[0])
        fn = self._lib.gridhip_grid_dev if dev else self._lib.gridhip_grid
        self._check(fn(self._h, a.shape[0], a.shape[1], self._ptr(a), n, self._ptr(pu), self._ptr(pv), stride,
                       self._ptr(vis)))
        return a

    def convgrid(self, gcf, a, p, v):
        """src/Gridding.hs:153-197 ; gcf [Q,Q,gh,gw]"""
        dev, a, pu, pv, stride, vis, _, gcf = self._prep(a, p, v, None, gcf)
        Q, Q2, gh, gw = gcf.shape
        assert Q == Q2
        n = int(pu.shape[0])
        fn = self._lib.gridhip_convgrid_dev if dev else self._lib.gridhip_convgrid
        self._check(fn(self._h, a.shape[0], a.shape[1], self._ptr(a), n, Q, gh, gw, self._ptr(gcf), self._ptr(pu),
                       self._ptr(pv), stride, self._ptr(vis)))
        return a

    def convgrid2(self, gcf, a, p, wbin, v):
        """src/Gridding.hs:199-244 ; gcf [W,Q,Q,gh,gw]"""
        dev, a, pu, pv, stride, vis, wbin, gcf = self._prep(a, p, v, wbin, gcf)
        W, Q, Q2, gh, gw = gcf.shape
        assert Q == Q2
        n = int(pu.shape[0])
        fn = self._lib.gridhip_convgrid2_dev if dev else self._lib.gridhip_convgrid2
        self._check(fn(self._h, a.shape[0], a.shape[1], self._ptr(a), n, W, Q, gh, gw, self._ptr(gcf),
                       self._ptr(pu), self._ptr(pv), stride, self._ptr(wbin), self._ptr(vis)))
        return a

    def degrid2(self, gcf, a, p, wbin, out=None):
        """Gather with convgrid2's coordinates: out[k] = sum_ij gcf[wbin,yf,xf,i,j] * a[y0+i,x0+j]."""
        dev, a, pu, pv, stride, _, wbin, gcf = self._prep(a, p, None, wbin, gcf)
        W, Q, Q2, gh, gw = gcf.shape
        assert Q == Q2
        n = int(pu.shape[0])
        if dev:
            import torch
            if out is None:
                out = torch.empty(n, dtype=torch.complex128, device=a.device)
        elif out is None:
            out = np.empty(n, dtype=np.complex128)
        fn = self._lib.gridhip_degrid2_dev if dev else self._lib.gridhip_degrid2
        self._check(fn(self._h, a.shape[0], a.shape[1], self._ptr(a), n, W, Q, gh, gw, self._ptr(gcf),
                       self._ptr(pu), self._ptr(pv), stride, self._ptr(wbin), self._ptr(out)))
        return out


    def plan(self, grid_shape, gcf_shape, p, wbin):
        """Bin the baselines `p` (torch cuda tensors) once for an [H, W] grid and a [W,Q,Q,gh,gw] kernel
        table; returns a Plan whose grid()/degrid() skip the pre-pass."""
        import torch
        u, v, stride = _split_p(p)
        if stride == 1:
            u, v = u.to(torch.float64).contiguous(), v.to(torch.float64).contiguous()
        wbin = None if wbin is None else wbin.to(torch.int64).contiguous()
        self._use_torch_stream()
        W, Q, _, gh, gw = gcf_shape
        h = C.c_void_p()
        self._check(self._lib.gridhip_plan_create_dev(self._h, grid_shape[0], grid_shape[1], int(u.shape[0]), W, Q, gh,
                                                      gw, self._ptr(u), self._ptr(v), stride, self._ptr(wbin),
                                                      C.byref(h)))
        return Plan(self, h, int(u.shape[0]), tuple(grid_shape), tuple(gcf_shape))

    def convgrid4(self, wkerns, akerns, a, p, index, v):
        """src/Gridding.hs:318-396 ; index = (wbin, a1, a2) arrays.  convgrid3 (:246-317) gives the same grid."""
        wbin, a1, a2 = index
        dev, a, pu, pv, stride, vis, wbin, wkerns = self._prep(a, p, v, wbin, wkerns)
        W, Q, _, S, _ = wkerns.shape
        n = int(pu.shape[0])
        if dev:
            import torch
            cv = lambda t, dt: t if (t.dtype == dt and t.is_contiguous()) else t.to(dt).contiguous()
            akerns, a1, a2 = cv(akerns, torch.complex128), cv(a1, torch.int64), cv(a2, torch.int64)
            fn = self._lib.gridhip_awgrid_dev
        else:
            akerns, a1, a2 = self._np(akerns, np.complex128), self._np(a1, np.int64), self._np(a2, np.int64)
            fn = self._lib.gridhip_awgrid
        self._check(fn(self._h, a.shape[0], a.shape[1], self._ptr(a), n, W, Q, S, akerns.shape[0], self._ptr(wkerns),
                       self._ptr(akerns), self._ptr(pu), self._ptr(pv), stride, self._ptr(wbin), self._ptr(a1),
                       self._ptr(a2), self._ptr(vis)))
        return a

    convgrid3 = convgrid4

    def aw_stats(self, S=15):
        """What the last convgrid4 did: visibilities keyed, distinct kernels built, the hit rate of the per-key
        de-duplication, the fp64 flops of the kernel builds and (timing enabled) their device time."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.gridhip_aw_last_stats(self._h, C.byref(a), C.byref(b)))
        c = S // 2
        macs = sum(S - abs(y - c) for y in range(S)) ** 2  # complex products of one 'same' S x S convolution (15: 28 561)
        out = {"vis_keyed": a.value, "kernels_built": b.value,
               "hit_rate": 1.0 - b.value / a.value if a.value else 0.0,
               "conv_flops_per_call": 8.0 * macs * b.value}
        try:
            t = self.timing(0)
            out.update(build_ms=t[1], grid_ms=t[2])
        except GridHipError:
            pass
        return out

    def aw_imaging(self, theta, lam, wkernels, wbins, akernels, uvw, src, vis):
        """src/Gridding.hs:452-478 (aw_imagingOld :480-506 gives the same grid); src = (a1, a2, t, f)"""
        u, v, w, st = self._uvw(uvw)
        vis = self._np(vis, np.complex128)
        wk, ak = self._np(wkernels, np.complex128), self._np(akernels, np.complex128)
        wv = self._np(wbins, np.float64)
        a1, a2 = self._np(src[0], np.int64), self._np(src[1], np.int64)
        W, Q, _, S, _ = wk.shape
        N = self.image_size(theta, lam)
        g = np.empty((N, N), dtype=np.complex128)
        self._check(self._lib.gridhip_aw_imaging(self._h, float(theta), int(lam), W, Q, S, ak.shape[0], self._ptr(wk),
                                                 self._ptr(wv), self._ptr(ak), len(vis), self._ptr(u), self._ptr(v),
                                                 self._ptr(w), st, self._ptr(a1), self._ptr(a2), self._ptr(vis),
                                                 self._ptr(g)))
        return g

    aw_imagingOld = aw_imaging

    # -- callers either side of the gridder (host arrays; src/Gridding.hs names) -----------------
    def image_size(self, theta, lam):
        return int(self._lib.gridhip_image_size(float(theta), int(lam)))

    def wbins(self, w, wstep):
        """w-bin rule of w_cache_imaging (:426-432) -> (wbin, wmin, nplanes)"""
        w = self._np(w, np.float64)
        out = np.empty(len(w), dtype=np.int64)
        mn, npl = C.c_int64(), C.c_int64()
        self._check(self._lib.gridhip_wbins(self._h, len(w), self._ptr(w), int(wstep), self._ptr(out), C.byref(mn),
                                            C.byref(npl)))
        return out, mn.value, npl.value

    def findClosest(self, ws, w):
        """:895-907, vectorised over w"""
        ws, w = self._np(ws, np.float64), self._np(np.atleast_1d(w), np.float64)
        out = np.empty(len(w), dtype=np.int64)
        self._check(self._lib.gridhip_find_closest(self._h, len(ws), self._ptr(ws), len(w), self._ptr(w),
                                                   self._ptr(out)))
        return out

    def mirror_uvw(self, uvw, vis):
        """:551-562 -> ((u, v, w), vis)"""
        u, v, w = (self._np(x, np.float64).copy() for x in uvw)
        vis = self._np(vis, np.complex128).copy()
        self._check(self._lib.gridhip_mirror_uvw(self._h, len(u), self._ptr(u), self._ptr(v), self._ptr(w),
                                                 self._ptr(vis)))
        return (u, v, w), vis

    def doweight(self, theta, lam, p, v):
        """:564-583 ; p in wavelengths"""
        u, vv = self._np(p[0], np.float64), self._np(p[1], np.float64)
        vis = self._np(v, np.complex128).copy()
        self._check(self._lib.gridhip_doweight(self._h, float(theta), int(lam), len(u), self._ptr(u), self._ptr(vv),
                                               self._ptr(vis)))
        return vis

    def make_grid_hermitian(self, guv):
        """:585-605"""
        g = self._np(guv, np.complex128).copy()
        self._check(self._lib.gridhip_make_grid_hermitian(self._h, g.shape[0], self._ptr(g)))
        return g

    def _fft(self, m, inverse):
        m = self._np(m, np.complex128)
        out = np.empty_like(m)
        self._check(self._lib.gridhip_fft2_centered(self._h, m.shape[0], self._ptr(m), self._ptr(out), int(inverse)))
        return out

    def ifft(self, m):
        """:828-829"""
        return self._fft(m, True)

    def fft(self, m):
        """:815-816 (fftO)"""
        return self._fft(m, False)

    def w_kernel(self, theta, w, npixFF, npixKern, qpx):
        """:610-619 -> [qpx, qpx, npixKern, npixKern]"""
        out = np.empty((qpx, qpx, npixKern, npixKern), dtype=np.complex128)
        self._check(self._lib.gridhip_w_kernel(self._h, float(theta), float(w), int(npixFF), int(npixKern), int(qpx),
                                               self._ptr(out)))
        return out

    def _uvw(self, uvw):
        if isinstance(uvw, (tuple, list)):
            u, v, w = (self._np(x, np.float64) for x in uvw)
            return u, v, w, 1
        m = self._np(uvw, np.float64)
        return m[:, 0], m[:, 1], m[:, 2], 3

    def simple_imaging(self, theta, lam, uvw, src, vis):
        """:84-93 (src is unused by this imaging function, as in the reference)"""
        u, v, _, st = self._uvw(uvw)
        vis = self._np(vis, np.complex128)
        N = self.image_size(theta, lam)
        g = np.empty((N, N), dtype=np.complex128)
        self._check(self._lib.gridhip_simple_imaging(self._h, float(theta), int(lam), len(vis), self._ptr(u),
                                                     self._ptr(v), st, self._ptr(vis), self._ptr(g)))
        return g

    def conv_imaging(self, kv, theta, lam, uvw, src, vis):
        """:115-124 ; kv [Q,Q,gh,gw]"""
        u, v, _, st = self._uvw(uvw)
        vis, kv = self._np(vis, np.complex128), self._np(kv, np.complex128)
        Q, _, gh, gw = kv.shape
        N = self.image_size(theta, lam)
        g = np.empty((N, N), dtype=np.complex128)
        self._check(self._lib.gridhip_conv_imaging(self._h, Q, gh, gw, self._ptr(kv), float(theta), int(lam), len(vis),
                                                   self._ptr(u), self._ptr(v), st, self._ptr(vis), self._ptr(g)))
        return g

    def _uvw_dev(self, uvw):
        """device-resident baselines: a (u, v, w) tuple of cuda float64 tensors or an (n, 3) cuda tensor"""
        import torch
        if isinstance(uvw, (tuple, list)):
            u, v, w = (x.to(torch.float64).contiguous() for x in uvw)
            return u, v, w, 1
        m = uvw.to(torch.float64).contiguous()
        return m[:, 0], m[:, 1], m[:, 2], 3

    def w_cache_imaging(self, kernops, theta, lam, uvw, src, vis):
        """:399-449 ; kernops = dict(wstep=, qpx=, npixFF=, npixKern=) as KernelOptions (:30-38).
        torch cuda tensors take the device-resident form (gridhip_w_cache_imaging_dev) and return a cuda tensor."""
        if _is_torch(vis):
            import torch
            u, v, w, st = self._uvw_dev(uvw)
            vis = vis.to(torch.complex128).contiguous()
            N = self.image_size(theta, lam)
            g = torch.empty((N, N), dtype=torch.complex128, device=vis.device)
            self._use_torch_stream()
            self._check(self._lib.gridhip_w_cache_imaging_dev(
                self._h, int(kernops.get("wstep") or 2000), int(kernops["qpx"]), int(kernops["npixFF"]),
                int(kernops["npixKern"]), float(theta), int(lam), int(vis.shape[0]), self._ptr(u), self._ptr(v),
                self._ptr(w), st, self._ptr(vis), self._ptr(g)))
            return g
        u, v, w, st = self._uvw(uvw)
        vis = self._np(vis, np.complex128)
        N = self.image_size(theta, lam)
        g = np.empty((N, N), dtype=np.complex128)
        self._check(self._lib.gridhip_w_cache_imaging(
            self._h, int(kernops.get("wstep") or 2000), int(kernops["qpx"]), int(kernops["npixFF"]),
            int(kernops["npixKern"]), float(theta), int(lam), len(vis), self._ptr(u), self._ptr(v), self._ptr(w), st,
            self._ptr(vis), self._ptr(g)))
        return g

    def do_imaging(self, theta, lam, uvw, a1, a2, t, f, vis, imgfn):
        """:509-549 -> (image, psf, pmax).  imgfn = ("simple",) | ("conv", kv) | ("w_cache", kernops);
        a1, a2, t, f (src) are accepted for signature parity and unused by these imaging functions.
        torch cuda tensors (uvw, vis, and kv for "conv") take the device-resident form (gridhip_do_imaging_dev):
        nothing crosses PCIe, image and psf come back as cuda tensors."""
        dev = _is_torch(vis)
        N = self.image_size(theta, lam)
        if dev:
            import torch
            u, v, w, st = self._uvw_dev(uvw)
            vis = vis.to(torch.complex128).contiguous()
            img = torch.empty((N, N), dtype=torch.float64, device=vis.device)
            psf = torch.empty((N, N), dtype=torch.float64, device=vis.device)
            self._use_torch_stream()
        else:
            u, v, w, st = self._uvw(uvw)
            vis = self._np(vis, np.complex128)
            img = np.empty((N, N), dtype=np.float64)
            psf = np.empty((N, N), dtype=np.float64)
        pmax = C.c_double()
        kind, wstep, Q, npixFF, gh, gw, kv = 0, 0, 0, 0, 0, 0, None
        if imgfn[0] == "conv":
            kv = imgfn[1].to(vis.dtype).contiguous() if dev else self._np(imgfn[1], np.complex128)
            kind, (Q, _, gh, gw) = 1, kv.shape
        elif imgfn[0] == "w_cache":
            ko = imgfn[1]
            kind, wstep, Q, npixFF, gh = 2, int(ko.get("wstep") or 2000), int(ko["qpx"]), int(ko["npixFF"]), int(ko["npixKern"])
            gw = gh
        elif imgfn[0] != "simple":
            raise ValueError("unknown imaging function")
        fn = self._lib.gridhip_do_imaging_dev if dev else self._lib.gridhip_do_imaging
        self._check(fn(self._h, kind, wstep, Q, npixFF, gh, gw, self._ptr(kv), float(theta), int(lam), int(vis.shape[0]),
                       self._ptr(u), self._ptr(v), self._ptr(w), st, self._ptr(vis), self._ptr(img), self._ptr(psf),
                       C.byref(pmax)))
        return img, psf, pmax.value


class Plan:
    """Baselines binned once (gridhip_plan); grid()/degrid() run the tile kernel only."""

    def __init__(self, ctx, handle, n, grid_shape, gcf_shape):
        self.ctx, self._h, self.n, self.grid_shape, self.gcf_shape = ctx, handle, n, grid_shape, gcf_shape

    def _chk(self, gcf, a):
        assert tuple(gcf.shape) == self.gcf_shape and tuple(a.shape) == self.grid_shape
        assert gcf.is_cuda and a.is_cuda and gcf.is_contiguous() and a.is_contiguous()
        self.ctx._use_torch_stream()

    def grid(self, gcf, a, v):
        """a += convgrid2 contributions of visibilities v (cuda complex128, length n)"""
        self._chk(gcf, a)
        assert v.shape[0] == self.n and v.is_contiguous()
        self.ctx._check(self.ctx._lib.gridhip_plan_grid_dev(self._h, Context._ptr(gcf), Context._ptr(v),
                                                            Context._ptr(a)))
        return a

    def degrid(self, gcf, a, out=None):
        import torch
        self._chk(gcf, a)
        if out is None:
            out = torch.empty(self.n, dtype=torch.complex128, device=a.device)
        self.ctx._check(self.ctx._lib.gridhip_plan_degrid_dev(self._h, Context._ptr(gcf), Context._ptr(a),
                                                              Context._ptr(out)))
        return out

    def close(self):
        if self._h:
            self.ctx._lib.gridhip_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def default_context(device=0):
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]


def _ctx_for(a):
    if _is_torch(a):
        return default_context(a.device.index or 0)
    return default_context(0)


def grid(a, p, v):
    return _ctx_for(a).grid(a, p, v)


def convgrid(gcf, a, p, v):
    return _ctx_for(a).convgrid(gcf, a, p, v)


def convgrid2(gcf, a, p, wbin, v):
    return _ctx_for(a).convgrid2(gcf, a, p, wbin, v)


def degrid2(gcf, a, p, wbin):
    return _ctx_for(a).degrid2(gcf, a, p, wbin)
