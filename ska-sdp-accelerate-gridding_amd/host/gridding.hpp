// gridding.hpp — C++ host-side mirror of the reference's gridding interface over the C ABI.
//
// The reference's host language is Haskell (src/Gridding.hs); its toolchain is absent from the
// build image, so the host side above include/gridhip.h is offered in C++ (header-only) with the
// reference's names, argument order and meaning:
//
//   grid a p v                      src/Gridding.hs:95-98
//   convgrid gcf a p v              :153-157
//   convgrid2 gcf a p wbin v        :199-204
//   convgrid3/convgrid4 wkerns akerns a p index v   :246-252, :318-324
//   simple_imaging / conv_imaging / w_cache_imaging / aw_imaging   :84, :115, :399, :452
//   do_imaging theta lam uvw a1 a2 t f vis imgfn    :509-519
//   mirror_uvw, doweight, make_grid_hermitian, ifft, w_kernel, findClosest
//
// Error behaviour: the reference's functions are total on well-formed input and `error` otherwise;
// here every C-ABI failure is thrown as gridding::Error carrying gridhip_last_error().
// Arrays are std::vector in the layouts of SURVEY.md §8b (complex = std::complex<double>, grids
// row-major [y][x]); a Matrix carries its shape.
#pragma once
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/gridhip.h"

namespace gridding {

using F = double;                        // src/Types.hs:7
using Visibility = std::complex<double>; // src/Types.hs:16
using Int = int64_t;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

template <typename T>
struct Matrix {  // row-major [h][w]
    Int h = 0, w = 0;
    std::vector<T> data;
    Matrix() = default;
    Matrix(Int h_, Int w_, T fill = T()) : h(h_), w(w_), data((size_t)h_ * w_, fill) {}
    T &operator()(Int y, Int x) { return data[(size_t)y * w + x]; }
    const T &operator()(Int y, Int x) const { return data[(size_t)y * w + x]; }
};

struct BaseLines {  // Vector (F,F,F) as Accelerate stores it: three arrays
    std::vector<F> u, v, w;
    size_t size() const { return u.size(); }
};

struct Kernel {  // [Q][Q][gh][gw]
    Int Q = 0, gh = 0, gw = 0;
    std::vector<Visibility> data;
};
struct WKernels {  // [W][Q][Q][gh][gw]
    Int W = 0, Q = 0, gh = 0, gw = 0;
    std::vector<Visibility> data;
};
struct AKernels {  // [A][S][S]
    Int A = 0, S = 0;
    std::vector<Visibility> data;
};
struct KernelOptions {  // src/Gridding.hs:30-38 (the fields the w-kernel path reads)
    Int wstep = 2000, qpx = 0, npixFF = 0, npixKern = 0;
};

class Backend {  // plays the role of the (run, runN) pair of `Runners`, src/Gridding.hs:28
   public:
    explicit Backend(int device = 0)
    {
        int rc = gridhip_create(device, &ctx_);
        if (rc) throw Error(rc, std::string("gridhip_create: ") + gridhip_strerror(rc));
    }
    ~Backend() { gridhip_destroy(ctx_); }
    Backend(const Backend &) = delete;
    Backend &operator=(const Backend &) = delete;
    gridhip_ctx *raw() { return ctx_; }

    static const double *cd(const std::vector<Visibility> &v) { return reinterpret_cast<const double *>(v.data()); }
    static double *cd(std::vector<Visibility> &v) { return reinterpret_cast<double *>(v.data()); }
    void check(int rc) const
    {
        if (rc) throw Error(rc, gridhip_last_error(ctx_));
    }

    // ---- gridders: the destination grid `a` is copied, accumulated into and returned ----
    Matrix<Visibility> grid(Matrix<Visibility> a, const BaseLines &p, const std::vector<Visibility> &v)
    {
        check(gridhip_grid(ctx_, a.h, a.w, cd(a.data), (Int)v.size(), p.u.data(), p.v.data(), 1, cd(v)));
        return a;
    }
    Matrix<Visibility> convgrid(const Kernel &gcf, Matrix<Visibility> a, const BaseLines &p,
                                const std::vector<Visibility> &v)
    {
        check(gridhip_convgrid(ctx_, a.h, a.w, cd(a.data), (Int)v.size(), gcf.Q, gcf.gh, gcf.gw, cd(gcf.data),
                               p.u.data(), p.v.data(), 1, cd(v)));
        return a;
    }
    Matrix<Visibility> convgrid2(const WKernels &gcf, Matrix<Visibility> a, const BaseLines &p,
                                 const std::vector<Int> &wbin, const std::vector<Visibility> &v)
    {
        check(gridhip_convgrid2(ctx_, a.h, a.w, cd(a.data), (Int)v.size(), gcf.W, gcf.Q, gcf.gh, gcf.gw,
                                cd(gcf.data), p.u.data(), p.v.data(), 1, wbin.data(), cd(v)));
        return a;
    }
    std::vector<Visibility> degrid2(const WKernels &gcf, const Matrix<Visibility> &a, const BaseLines &p,
                                    const std::vector<Int> &wbin)
    {
        std::vector<Visibility> out(p.size());
        check(gridhip_degrid2(ctx_, a.h, a.w, cd(a.data), (Int)p.size(), gcf.W, gcf.Q, gcf.gh, gcf.gw, cd(gcf.data),
                              p.u.data(), p.v.data(), 1, wbin.data(), cd(out)));
        return out;
    }
    // index = (wbin, a1, a2) as three arrays
    Matrix<Visibility> convgrid4(const WKernels &wk, const AKernels &ak, Matrix<Visibility> a, const BaseLines &p,
                                 const std::vector<Int> &wbin, const std::vector<Int> &a1,
                                 const std::vector<Int> &a2, const std::vector<Visibility> &v)
    {
        check(gridhip_awgrid(ctx_, a.h, a.w, cd(a.data), (Int)v.size(), wk.W, wk.Q, wk.gh, ak.A, cd(wk.data),
                             cd(ak.data), p.u.data(), p.v.data(), 1, wbin.data(), a1.data(), a2.data(), cd(v)));
        return a;
    }
    Matrix<Visibility> convgrid3(const WKernels &wk, const AKernels &ak, Matrix<Visibility> a, const BaseLines &p,
                                 const std::vector<Int> &wbin, const std::vector<Int> &a1,
                                 const std::vector<Int> &a2, const std::vector<Visibility> &v)
    {
        return convgrid4(wk, ak, std::move(a), p, wbin, a1, a2, v);
    }

    // ---- imaging functions (uvw in wavelengths) ----
    Matrix<Visibility> simple_imaging(F theta, Int lam, const BaseLines &uvw, const std::vector<Visibility> &vis)
    {
        const Int N = gridhip_image_size(theta, lam);
        Matrix<Visibility> g(N, N);
        check(gridhip_simple_imaging(ctx_, theta, lam, (Int)vis.size(), uvw.u.data(), uvw.v.data(), 1, cd(vis),
                                     cd(g.data)));
        return g;
    }
    Matrix<Visibility> conv_imaging(const Kernel &kv, F theta, Int lam, const BaseLines &uvw,
                                    const std::vector<Visibility> &vis)
    {
        const Int N = gridhip_image_size(theta, lam);
        Matrix<Visibility> g(N, N);
        check(gridhip_conv_imaging(ctx_, kv.Q, kv.gh, kv.gw, cd(kv.data), theta, lam, (Int)vis.size(), uvw.u.data(),
                                   uvw.v.data(), 1, cd(vis), cd(g.data)));
        return g;
    }
    Matrix<Visibility> w_cache_imaging(const KernelOptions &ko, F theta, Int lam, const BaseLines &uvw,
                                       const std::vector<Visibility> &vis)
    {
        const Int N = gridhip_image_size(theta, lam);
        Matrix<Visibility> g(N, N);
        check(gridhip_w_cache_imaging(ctx_, ko.wstep, ko.qpx, ko.npixFF, ko.npixKern, theta, lam, (Int)vis.size(),
                                      uvw.u.data(), uvw.v.data(), uvw.w.data(), 1, cd(vis), cd(g.data)));
        return g;
    }
    Matrix<Visibility> aw_imaging(F theta, Int lam, const WKernels &wk, const std::vector<F> &wbins,
                                  const AKernels &ak, const BaseLines &uvw, const std::vector<Int> &a1,
                                  const std::vector<Int> &a2, const std::vector<Visibility> &vis)
    {
        const Int N = gridhip_image_size(theta, lam);
        Matrix<Visibility> g(N, N);
        check(gridhip_aw_imaging(ctx_, theta, lam, wk.W, wk.Q, wk.gh, ak.A, cd(wk.data), wbins.data(), cd(ak.data),
                                 (Int)vis.size(), uvw.u.data(), uvw.v.data(), uvw.w.data(), 1, a1.data(), a2.data(),
                                 cd(vis), cd(g.data)));
        return g;
    }

    // do_imaging with w_cache_imaging kernops as the imaging function -> (image, psf, pmax)
    std::tuple<Matrix<F>, Matrix<F>, F> do_imaging_w_cache(F theta, Int lam, const BaseLines &uvw,
                                                          const std::vector<Visibility> &vis,
                                                          const KernelOptions &ko)
    {
        const Int N = gridhip_image_size(theta, lam);
        Matrix<F> img(N, N), psf(N, N);
        F pmax = 0;
        check(gridhip_do_imaging(ctx_, 2, ko.wstep, ko.qpx, ko.npixFF, ko.npixKern, ko.npixKern, nullptr, theta, lam,
                                 (Int)vis.size(), uvw.u.data(), uvw.v.data(), uvw.w.data(), 1, cd(vis),
                                 img.data.data(), psf.data.data(), &pmax));
        return {std::move(img), std::move(psf), pmax};
    }
    std::tuple<Matrix<F>, Matrix<F>, F> do_imaging_simple(F theta, Int lam, const BaseLines &uvw,
                                                         const std::vector<Visibility> &vis)
    {
        const Int N = gridhip_image_size(theta, lam);
        Matrix<F> img(N, N), psf(N, N);
        F pmax = 0;
        check(gridhip_do_imaging(ctx_, 0, 0, 0, 0, 0, 0, nullptr, theta, lam, (Int)vis.size(), uvw.u.data(),
                                 uvw.v.data(), uvw.w.data(), 1, cd(vis), img.data.data(), psf.data.data(), &pmax));
        return {std::move(img), std::move(psf), pmax};
    }

    // ---- helpers ----
    Matrix<Visibility> make_grid_hermitian(Matrix<Visibility> g)
    {
        check(gridhip_make_grid_hermitian(ctx_, g.h, cd(g.data)));
        return g;
    }
    Matrix<Visibility> ifft(const Matrix<Visibility> &m)
    {
        Matrix<Visibility> out(m.h, m.w);
        check(gridhip_fft2_centered(ctx_, m.h, cd(m.data), cd(out.data), 1));
        return out;
    }
    Kernel w_kernel(F theta, F w, const KernelOptions &ko)
    {
        Kernel k;
        k.Q = ko.qpx;
        k.gh = k.gw = ko.npixKern;
        k.data.resize((size_t)k.Q * k.Q * k.gh * k.gw);
        check(gridhip_w_kernel(ctx_, theta, w, ko.npixFF, ko.npixKern, ko.qpx, cd(k.data)));
        return k;
    }
    std::vector<Int> findClosest(const std::vector<F> &ws, const std::vector<F> &w)
    {
        std::vector<Int> out(w.size());
        check(gridhip_find_closest(ctx_, (Int)ws.size(), ws.data(), (Int)w.size(), w.data(), out.data()));
        return out;
    }

   private:
    gridhip_ctx *ctx_ = nullptr;
};


// A whole node: visibility-sharded gridding over several GPUs of one process with one RCCL fp64 all-reduce of the
// partial grids (include/gridhip.h, gridhip_comm_*).  The reference has no counterpart (app/Main.hs:46-53 picks one
// (run, runN) pair); the signature stays convgrid2's.
class Node {
   public:
    explicit Node(int ndev, const int *dev_ids = nullptr)
    {
        int rc = gridhip_comm_create(ndev, dev_ids, &comm_);
        if (rc) throw Error(rc, std::string("gridhip_comm_create: ") + gridhip_comm_last_error(nullptr));
    }
    ~Node() { gridhip_comm_destroy(comm_); }
    Node(const Node &) = delete;
    Node &operator=(const Node &) = delete;
    int devices() const { return gridhip_comm_ndev(comm_); }

    Matrix<Visibility> convgrid2(const WKernels &gcf, Matrix<Visibility> a, const BaseLines &p,
                                 const std::vector<Int> &wbin, const std::vector<Visibility> &v)
    {
        int rc = gridhip_comm_convgrid2(comm_, a.h, a.w, reinterpret_cast<double *>(a.data.data()), (Int)v.size(), gcf.W,
                                        gcf.Q, gcf.gh, gcf.gw, reinterpret_cast<const double *>(gcf.data.data()),
                                        p.u.data(), p.v.data(), 1, wbin.data(), reinterpret_cast<const double *>(v.data()));
        if (rc) throw Error(rc, gridhip_comm_last_error(comm_));
        return a;
    }

   private:
    gridhip_comm *comm_ = nullptr;
};

}  // namespace gridding
