/*
 * gridref — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the gridding hot path of sakehl/SKA-SDP-Accelerate-gridding
 * (reference file: src/Gridding.hs).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; libgridhip.so never does.
 *
 * Parity status: the reference (Haskell + Accelerate) cannot be built or run here, its
 * data files are Git-LFS stubs, and its own tests record exactly ONE expected output
 * (old/BrokenNumbers.hs:86-91, the 5x5 `permute (+)` result).  That KAT pins the
 * accumulate/orientation semantics of this oracle (tests/test_oracle.py).  Everything
 * else — frac_coord, convgrid*, kernels, weighting, FFT — is "PARITY UNPINNED" by the
 * reference: this file is cross-checked against an independent numpy restatement
 * (oracle/gridref_np.py) and against derived KATs built from the literal inputs of the
 * reference's own test scripts (test/GridTesting.hs:389-426, test/SmallTest.hs:51-76).
 *
 * Conventions (SURVEY.md §8 preamble):
 *   - complex arrays are interleaved (re, im) doubles  (src/Hdf5.hs:113-137, hdf5/hdf5.cc:14-17)
 *   - grids are row-major [y][x], y <-> v axis         (src/Gridding.hs:106-109,185-188)
 *   - gcf  : [W][Q][Q][gh][gw] complex, index order (wbin, yf, xf, i(row), j(col))
 *   - Int = int64_t, F = double                         (src/Types.hs:7-16)
 */
#ifndef GRIDREF_H
#define GRIDREF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* src/Gridding.hs:126-140 */
void gridref_frac_coord(int64_t n, int64_t qpx, int64_t cnt, const double *p,
                        int64_t *flx, int64_t *fracx);
/* src/Gridding.hs:95-112 (out-of-range cells are dropped; the reference does not check) */
void gridref_grid(int64_t H, int64_t Wd, double *G, int64_t n, const double *u,
                  const double *v, const double *vis);
/* src/Gridding.hs:153-197 */
void gridref_convgrid(int64_t H, int64_t Wd, double *G, int64_t n, int64_t Q, int64_t gh,
                      int64_t gw, const double *gcf, const double *u, const double *v,
                      const double *vis);
/* src/Gridding.hs:199-244 */
void gridref_convgrid2(int64_t H, int64_t Wd, double *G, int64_t n, int64_t W, int64_t Q,
                       int64_t gh, int64_t gw, const double *gcf, const double *u,
                       const double *v, const int64_t *wbin, const double *vis);
/* OpenMP variants used only as the timed CPU baseline (bench.py cpu_baseline):
 * mode 0 = shared grid + atomic updates, mode 1 = per-thread private grids + reduce. */
void gridref_convgrid2_mt(int64_t H, int64_t Wd, double *G, int64_t n, int64_t W, int64_t Q,
                          int64_t gh, int64_t gw, const double *gcf, const double *u,
                          const double *v, const int64_t *wbin, const double *vis,
                          int mode, int nthreads);
int gridref_max_threads(void);
/* adjoint-pattern gather of convgrid2 (absent from the reference; SURVEY.md §8a "degrid") */
void gridref_degrid2(int64_t H, int64_t Wd, const double *G, int64_t n, int64_t W, int64_t Q,
                     int64_t gh, int64_t gw, const double *gcf, const double *u,
                     const double *v, const int64_t *wbin, double *vis_out);

/* src/Gridding.hs:895-907 (hi clamped to len-1) */
int64_t gridref_find_closest(int64_t nws, const double *ws, double w);
/* src/Gridding.hs:426-432 */
void gridref_wbins(int64_t n, const double *w, int64_t wstep, int64_t *wbin,
                   int64_t *wmin_out, int64_t *nplanes_out);
/* src/Gridding.hs:551-562 (in place) */
void gridref_mirror_uvw(int64_t n, double *u, double *v, double *w, double *vis);
/* src/Gridding.hs:564-583: p = uvw/lam already applied; vis (in/out) divided by cell count */
void gridref_doweight(int64_t N, int64_t n, const double *pu, const double *pv, double *vis);
/* src/Gridding.hs:585-605 (in place) */
void gridref_make_grid_hermitian(int64_t N, double *G);

/* src/Gridding.hs:795-811 incl. pad_mid :682-691, padder :863-877 (transpose quirk), extract_mid :694-707 */
void gridref_convolve2d(int64_t n, const double *a1, const double *a2, double *out);
/* algebraically identical direct form: same_conv(a1,a2)^T */
void gridref_convolve2d_direct(int64_t n, const double *a1, const double *a2, double *out);
/* src/Gridding.hs:761-775 ; wkern is one w-plane [Q][Q][S][S] */
void gridref_aw_kernel_fn2(int64_t Q, int64_t S, int64_t yf, int64_t xf, const double *wkern,
                           const double *a1, const double *a2, double *out, int direct);
/* src/Gridding.hs:246-396 — convgrid3 and convgrid4 produce the same grid */
void gridref_awgrid(int64_t H, int64_t Wd, double *G, int64_t n, int64_t W, int64_t Q,
                    int64_t S, int64_t A, const double *wkerns, const double *akerns,
                    const double *u, const double *v, const int64_t *wbin,
                    const int64_t *a1, const int64_t *a2, const double *vis, int direct);

/* src/Gridding.hs:610-728: out is [Q][Q][S][S] */
int gridref_w_kernel(double theta, double w, int64_t npixFF, int64_t npixKern, int64_t qpx,
                     double *out);
/* centred 2-D transforms, src/Gridding.hs:815-829 (inverse is 1/N^2 normalised) */
int gridref_fft2_centered(int64_t N, const double *in, double *out, int inverse);

#ifdef __cplusplus
}
#endif
#endif
