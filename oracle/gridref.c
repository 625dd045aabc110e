/*
 * gridref.c — CPU ORACLE (test infrastructure, NOT product code). See gridref.h.
 *
 * Every function cites the line range of /root/reference/src/Gridding.hs it restates.
 * Accumulation order is fixed (visibility-major, then row i, then column j) so that the
 * oracle itself is deterministic.  Compile with -ffp-contract=off: the index math of
 * frac_coord must not be fused (SURVEY.md §7 "Index parity").
 */
#include "gridref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static inline int64_t idiv2(int64_t a) { return a / 2; } /* `div` 2 on non-negative ints */

/* ---- frac_coord, src/Gridding.hs:126-140 -------------------------------------------
 *   x     = fromIntegral (n `div` 2) + p * fromIntegral n          (:137)
 *   flx   = floor (x + 0.5 / qpx)                                  (:138)
 *   fracx = round ((x - fromIntegral flx) * qpx)                   (:139)
 * `round` follows libm round() as the LLVM backends do; exact ties are
 * outside the parity contract.  fracx is clamped to [0, qpx-1]: the reference would read
 * the kernel out of range at index -1 / qpx for such a tie. */
static inline void frac_coord1(int64_t n, int64_t qpx, double p, int64_t *flx, int64_t *fr)
{
    double halfnf = (double)idiv2(n);
    double nf = (double)n;
    double qpxf = (double)qpx;
    double qpxfrac = 0.5 / qpxf;
    double x = halfnf + p * nf;
    double fl = floor(x + qpxfrac);
    int64_t f = (int64_t)fl;
    int64_t r = (int64_t)round((x - (double)f) * qpxf);
    if (r < 0) r = 0;
    if (r > qpx - 1) r = qpx - 1;
    *flx = f;
    *fr = r;
}

void gridref_frac_coord(int64_t n, int64_t qpx, int64_t cnt, const double *p,
                        int64_t *flx, int64_t *fracx)
{
    for (int64_t k = 0; k < cnt; ++k) frac_coord1(n, qpx, p[k], &flx[k], &fracx[k]);
}

/* ---- grid, src/Gridding.hs:95-112 ---------------------------------------------------
 *   toGridCell f = halfn + floor (0.5 + nf * f)   with n = height of the grid  (:101-112)
 *   G[toGridCell v, toGridCell u] += vis          (index2 y x, :109)
 * The reference performs no bounds check here; an out-of-range cell is undefined there
 * and dropped here. */
void gridref_grid(int64_t H, int64_t Wd, double *G, int64_t n, const double *u,
                  const double *v, const double *vis)
{
    int64_t halfn = idiv2(H);
    double nf = (double)H;
    for (int64_t k = 0; k < n; ++k) {
        int64_t x = halfn + (int64_t)floor(0.5 + nf * u[k]);
        int64_t y = halfn + (int64_t)floor(0.5 + nf * v[k]);
        if (x < 0 || y < 0 || x >= Wd || y >= H) continue;
        G[2 * (y * Wd + x)] += vis[2 * k];
        G[2 * (y * Wd + x) + 1] += vis[2 * k + 1];
    }
}

/* ---- convgrid2, src/Gridding.hs:199-244 (convgrid :153-197 is the W=1 case) ---------
 *   (x,xf) = frac_coord width  qpx u ; (y,yf) = frac_coord height qpx v   (:142-151,:212)
 *   x0 = x - gw `div` 2 ; y0 = y - gh `div` 2                             (:217-218)
 *   for i<gh, j<gw: (xx,yy) = (x0+j, y0+i)                                (:241-242)
 *       fixoutofbounds: out of range -> add 0 to G[0,0], i.e. dropped     (:883-891)
 *       G[yy,xx] += vis * gcf[wbin,yf,xf,i,j]                             (:243-244)
 * complex product as Data.Complex: (a:+b)*(c:+d) = (ac-bd) :+ (ad+bc). */
static inline void conv_one(int64_t H, int64_t Wd, double *G, int64_t Q, int64_t gh,
                            int64_t gw, const double *gcf, double pu, double pv,
                            int64_t wb, double vr, double vi)
{
    int64_t x, xf, y, yf;
    frac_coord1(Wd, Q, pu, &x, &xf);
    frac_coord1(H, Q, pv, &y, &yf);
    int64_t x0 = x - idiv2(gw), y0 = y - idiv2(gh);
    const double *k = gcf + 2 * ((((wb * Q) + yf) * Q + xf) * gh * gw);
    for (int64_t i = 0; i < gh; ++i) {
        int64_t yy = y0 + i;
        for (int64_t j = 0; j < gw; ++j) {
            int64_t xx = x0 + j;
            if (xx < 0 || yy < 0 || xx >= Wd || yy >= H) continue;
            double kr = k[2 * (i * gw + j)], ki = k[2 * (i * gw + j) + 1];
            double *g = G + 2 * (yy * Wd + xx);
            g[0] += vr * kr - vi * ki;
            g[1] += vr * ki + vi * kr;
        }
    }
}

void gridref_convgrid2(int64_t H, int64_t Wd, double *G, int64_t n, int64_t W, int64_t Q,
                       int64_t gh, int64_t gw, const double *gcf, const double *u,
                       const double *v, const int64_t *wbin, const double *vis)
{
    (void)W;
    for (int64_t k = 0; k < n; ++k)
        conv_one(H, Wd, G, Q, gh, gw, gcf, u[k], v[k], wbin ? wbin[k] : 0, vis[2 * k],
                 vis[2 * k + 1]);
}

void gridref_convgrid(int64_t H, int64_t Wd, double *G, int64_t n, int64_t Q, int64_t gh,
                      int64_t gw, const double *gcf, const double *u, const double *v,
                      const double *vis)
{
    gridref_convgrid2(H, Wd, G, n, 1, Q, gh, gw, gcf, u, v, NULL, vis);
}

int gridref_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Timed CPU baseline only (bench.py cpu_baseline).  mode 0 mirrors what a parallel
 * `permute (+)` does (shared destination, atomic element updates); mode 1 keeps private grids
 * and reduces them; mode 2 (the strongest) lets every band of grid rows be updated by one thread. */
void gridref_convgrid2_mt(int64_t H, int64_t Wd, double *G, int64_t n, int64_t W, int64_t Q,
                          int64_t gh, int64_t gw, const double *gcf, const double *u,
                          const double *v, const int64_t *wbin, const double *vis,
                          int mode, int nthreads)
{
    (void)W;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    if (mode == 0) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int64_t k = 0; k < n; ++k) {
            int64_t x, xf, y, yf;
            frac_coord1(Wd, Q, u[k], &x, &xf);
            frac_coord1(H, Q, v[k], &y, &yf);
            int64_t x0 = x - idiv2(gw), y0 = y - idiv2(gh);
            int64_t wb = wbin ? wbin[k] : 0;
            const double *kk = gcf + 2 * ((((wb * Q) + yf) * Q + xf) * gh * gw);
            double vr = vis[2 * k], vi = vis[2 * k + 1];
            for (int64_t i = 0; i < gh; ++i) {
                int64_t yy = y0 + i;
                for (int64_t j = 0; j < gw; ++j) {
                    int64_t xx = x0 + j;
                    if (xx < 0 || yy < 0 || xx >= Wd || yy >= H) continue;
                    double kr = kk[2 * (i * gw + j)], ki = kk[2 * (i * gw + j) + 1];
                    double *g = G + 2 * (yy * Wd + xx);
                    double ar = vr * kr - vi * ki, ai = vr * ki + vi * kr;
#pragma omp atomic
                    g[0] += ar;
#pragma omp atomic
                    g[1] += ai;
                }
            }
        }
    } else if (mode == 2) {
        /* owner computes: the grid is cut into bands of rows; a band is updated by one thread only, which scans
         * the footprint rows of every visibility and applies those that fall into its band.  No atomics, no
         * private grids, and every cell still receives its contributions in visibility order (bit-identical to
         * the serial oracle).  Bands are handed out dynamically (mirrored data fills only half the grid).
         * n < 2^31 (indices are kept as int32). */
        int32_t *y0s = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
        if (!y0s) {
            gridref_convgrid2(H, Wd, G, n, W, Q, gh, gw, gcf, u, v, wbin, vis);
            return;
        }
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int64_t k = 0; k < n; ++k) {
            int64_t y, yf;
            frac_coord1(H, Q, v[k], &y, &yf);
            int64_t y0 = y - idiv2(gh);
            if (!(v[k] == v[k]) || y0 < -(int64_t)1000000000 || y0 > (int64_t)1000000000) y0 = 1000000000; /* outside */
            y0s[k] = (int32_t)y0;
        }
        /* every visibility is listed under the one or two bands its footprint rows touch (band height >= gh), in
         * visibility order: per-thread counts -> offsets -> fill */
        int64_t band = gh > 32 ? gh : 32;
        int64_t nbands = (H + band - 1) / band;
        int64_t *cnt = (int64_t *)calloc((size_t)(nbands + 1) * (size_t)nthreads + 1, sizeof(int64_t));
        int64_t *start = (int64_t *)malloc((size_t)(nbands + 1) * sizeof(int64_t));
        int32_t *list = NULL;
        if (cnt && start) {
#pragma omp parallel num_threads(nthreads)
            {
                int t = omp_get_thread_num(), nt = omp_get_num_threads();
                int64_t k0 = n * t / nt, k1 = n * (t + 1) / nt;
                int64_t *mine = cnt + (size_t)t * (size_t)(nbands + 1);
                for (int64_t k = k0; k < k1; ++k) {
                    int64_t y0 = y0s[k];
                    if (y0 >= H || y0 + gh <= 0) continue;
                    int64_t b0 = y0 < 0 ? 0 : y0 / band, b1 = (y0 + gh - 1) / band;
                    if (b1 >= nbands) b1 = nbands - 1;
                    for (int64_t b = b0; b <= b1; ++b) ++mine[b];
                }
            }
            int64_t run = 0;
            for (int64_t b = 0; b < nbands; ++b) {
                start[b] = run;
                for (int t = 0; t < nthreads; ++t) {
                    int64_t c = cnt[(size_t)t * (size_t)(nbands + 1) + (size_t)b];
                    cnt[(size_t)t * (size_t)(nbands + 1) + (size_t)b] = run;
                    run += c;
                }
            }
            start[nbands] = run;
            list = (int32_t *)malloc((size_t)(run > 0 ? run : 1) * sizeof(int32_t));
        }
        if (!list) {
            free(cnt); free(start); free(y0s);
            gridref_convgrid2(H, Wd, G, n, W, Q, gh, gw, gcf, u, v, wbin, vis);
            return;
        }
#pragma omp parallel num_threads(nthreads)
        {
            int t = omp_get_thread_num(), nt = omp_get_num_threads();
            int64_t k0 = n * t / nt, k1 = n * (t + 1) / nt;
            int64_t *mine = cnt + (size_t)t * (size_t)(nbands + 1);
            for (int64_t k = k0; k < k1; ++k) {
                int64_t y0 = y0s[k];
                if (y0 >= H || y0 + gh <= 0) continue;
                int64_t b0 = y0 < 0 ? 0 : y0 / band, b1 = (y0 + gh - 1) / band;
                if (b1 >= nbands) b1 = nbands - 1;
                for (int64_t b = b0; b <= b1; ++b) list[mine[b]++] = (int32_t)k;
            }
        }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (int64_t b = 0; b < nbands; ++b) {
            int64_t r0 = b * band, r1 = r0 + band < H ? r0 + band : H;
            for (int64_t q = start[b]; q < start[b + 1]; ++q) {
                int64_t k = list[q];
                int64_t y0 = y0s[k];
                int64_t x, xf, y, yf;
                frac_coord1(Wd, Q, u[k], &x, &xf);
                frac_coord1(H, Q, v[k], &y, &yf);
                int64_t x0 = x - idiv2(gw);
                int64_t wb = wbin ? wbin[k] : 0;
                const double *kk = gcf + 2 * ((((wb * Q) + yf) * Q + xf) * gh * gw);
                double vr = vis[2 * k], vi = vis[2 * k + 1];
                int64_t i0 = r0 > y0 ? r0 - y0 : 0, i1 = r1 - y0 < gh ? r1 - y0 : gh;
                for (int64_t i = i0; i < i1; ++i) {
                    int64_t yy = y0 + i;
                    for (int64_t j = 0; j < gw; ++j) {
                        int64_t xx = x0 + j;
                        if (xx < 0 || xx >= Wd) continue;
                        double kr = kk[2 * (i * gw + j)], ki = kk[2 * (i * gw + j) + 1];
                        double *g = G + 2 * (yy * Wd + xx);
                        g[0] += vr * kr - vi * ki;
                        g[1] += vr * ki + vi * kr;
                    }
                }
            }
        }
        free(list); free(cnt); free(start);
        free(y0s);
    } else {
        size_t cells = (size_t)H * (size_t)Wd * 2;
        double *priv = (double *)calloc(cells * (size_t)nthreads, sizeof(double));
        if (!priv) { /* fall back to serial */
            gridref_convgrid2(H, Wd, G, n, W, Q, gh, gw, gcf, u, v, wbin, vis);
            return;
        }
#pragma omp parallel num_threads(nthreads)
        {
            int t = omp_get_thread_num();
            double *mine = priv + cells * (size_t)t;
#pragma omp for schedule(static)
            for (int64_t k = 0; k < n; ++k)
                conv_one(H, Wd, mine, Q, gh, gw, gcf, u[k], v[k], wbin ? wbin[k] : 0,
                         vis[2 * k], vis[2 * k + 1]);
#pragma omp for schedule(static)
            for (int64_t c = 0; c < (int64_t)cells; ++c) {
                double s = 0.0;
                for (int tt = 0; tt < nthreads; ++tt) s += priv[cells * (size_t)tt + (size_t)c];
                G[c] += s;
            }
        }
        free(priv);
    }
#else
    (void)mode; (void)nthreads;
    gridref_convgrid2(H, Wd, G, n, W, Q, gh, gw, gcf, u, v, wbin, vis);
#endif
}

/* ---- degrid2 (not in the reference) ----------------------------------
 *   out[k] = sum_{i,j} gcf[wbin_k,yf_k,xf_k,i,j] * G[y0+i, x0+j], out-of-range taps give 0 */
void gridref_degrid2(int64_t H, int64_t Wd, const double *G, int64_t n, int64_t W, int64_t Q,
                     int64_t gh, int64_t gw, const double *gcf, const double *u,
                     const double *v, const int64_t *wbin, double *vis_out)
{
    (void)W;
    for (int64_t k = 0; k < n; ++k) {
        int64_t x, xf, y, yf;
        frac_coord1(Wd, Q, u[k], &x, &xf);
        frac_coord1(H, Q, v[k], &y, &yf);
        int64_t x0 = x - idiv2(gw), y0 = y - idiv2(gh);
        int64_t wb = wbin ? wbin[k] : 0;
        const double *kk = gcf + 2 * ((((wb * Q) + yf) * Q + xf) * gh * gw);
        double sr = 0.0, si = 0.0;
        for (int64_t i = 0; i < gh; ++i) {
            int64_t yy = y0 + i;
            for (int64_t j = 0; j < gw; ++j) {
                int64_t xx = x0 + j;
                if (xx < 0 || yy < 0 || xx >= Wd || yy >= H) continue;
                double kr = kk[2 * (i * gw + j)], ki = kk[2 * (i * gw + j) + 1];
                const double *g = G + 2 * (yy * Wd + xx);
                sr += kr * g[0] - ki * g[1];
                si += kr * g[1] + ki * g[0];
            }
        }
        vis_out[2 * k] = sr;
        vis_out[2 * k + 1] = si;
    }
}

/* ---- findClosest, src/Gridding.hs:895-907 --------------------------------------------
 *   (lo,hi) = (0,len); while (hi-lo) `div` 2 >= 1: mid=(hi+lo) `div` 2;
 *       if w > ws[mid] then lo=mid else hi=mid
 *   |w-ws[lo]| < |w-ws[hi]| ? lo : hi
 * `hi` starts at len (one past the end) in the reference; reads of ws[len] are clamped to
 * the last element here, as the host twin src/ImageDataset.hs:150-168 (max = len-1) does. */
int64_t gridref_find_closest(int64_t nws, const double *ws, double w)
{
    int64_t lo = 0, hi = nws;
    while ((hi - lo) / 2 >= 1) {
        int64_t mid = (hi + lo) / 2;
        if (w > ws[mid]) lo = mid; else hi = mid;
    }
    int64_t hc = hi > nws - 1 ? nws - 1 : hi;
    return fabs(w - ws[lo]) < fabs(w - ws[hc]) ? lo : hc;
}

/* ---- w-bin rule of w_cache_imaging, src/Gridding.hs:426-432 --------------------------
 *   roundedw = wstep * round (w / wstep) ; wbin = (roundedw - min roundedw) `div` wstep
 *   steps = (max - min) `div` wstep + 1 */
void gridref_wbins(int64_t n, const double *w, int64_t wstep, int64_t *wbin,
                   int64_t *wmin_out, int64_t *nplanes_out)
{
    int64_t mn = 0, mx = 0;
    for (int64_t k = 0; k < n; ++k) {
        int64_t rw = wstep * (int64_t)round(w[k] / (double)wstep);
        wbin[k] = rw;
        if (k == 0 || rw < mn) mn = rw;
        if (k == 0 || rw > mx) mx = rw;
    }
    for (int64_t k = 0; k < n; ++k) wbin[k] = (wbin[k] - mn) / wstep;
    if (wmin_out) *wmin_out = mn;
    if (nplanes_out) *nplanes_out = n ? (mx - mn) / wstep + 1 : 0;
}

/* ---- mirror_uvw, src/Gridding.hs:551-562 ---------------------------------------------- */
void gridref_mirror_uvw(int64_t n, double *u, double *v, double *w, double *vis)
{
    for (int64_t k = 0; k < n; ++k) {
        if (v[k] < 0) {
            u[k] = -u[k]; v[k] = -v[k]; w[k] = -w[k];
            vis[2 * k + 1] = -vis[2 * k + 1];
        }
    }
}

/* ---- doweight, src/Gridding.hs:564-583 --------------------------------------------------
 *   coords = frac_coords (N,N) 1 p ; weights = histogram of (y,x) ; v / weights[y,x]
 * (no bounds check in the reference; out-of-range visibilities are left untouched here) */
void gridref_doweight(int64_t N, int64_t n, const double *pu, const double *pv, double *vis)
{
    double *cnt = (double *)calloc((size_t)N * (size_t)N, sizeof(double));
    int64_t *cell = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t k = 0; k < n; ++k) {
        int64_t x, xf, y, yf;
        frac_coord1(N, 1, pu[k], &x, &xf);
        frac_coord1(N, 1, pv[k], &y, &yf);
        if (x < 0 || y < 0 || x >= N || y >= N) { cell[k] = -1; continue; }
        cell[k] = y * N + x;
        cnt[cell[k]] += 1.0;
    }
    for (int64_t k = 0; k < n; ++k) {
        if (cell[k] < 0) continue;
        vis[2 * k] /= cnt[cell[k]];
        vis[2 * k + 1] /= cnt[cell[k]];
    }
    free(cnt); free(cell);
}

/* ---- make_grid_hermitian, src/Gridding.hs:585-605 ---------------------------------------
 *   even N: G[y,x] += (x==0||y==0) ? 0 : conj(G[N-y,N-x])        (:589-594,:599,:602)
 *   odd  N: G[y,x] += conj(G[N-1-y,N-1-x])                        (:598,:601) */
void gridref_make_grid_hermitian(int64_t N, double *G)
{
    size_t cells = (size_t)N * (size_t)N;
    double *src = (double *)malloc(cells * 2 * sizeof(double));
    memcpy(src, G, cells * 2 * sizeof(double));
    int even = (N % 2) == 0;
    for (int64_t y = 0; y < N; ++y)
        for (int64_t x = 0; x < N; ++x) {
            double ar, ai;
            if (even) {
                if (x == 0 || y == 0) { ar = 0.0; ai = 0.0; }
                else {
                    const double *s = src + 2 * ((N - y) * N + (N - x));
                    ar = s[0]; ai = -s[1];
                }
            } else {
                const double *s = src + 2 * ((N - 1 - y) * N + (N - 1 - x));
                ar = s[0]; ai = -s[1];
            }
            G[2 * (y * N + x)] += ar;
            G[2 * (y * N + x) + 1] += ai;
        }
    free(src);
}

/* ===================== FFT (any length: radix-2, Bluestein otherwise) =================== */
static void fft_pow2(int64_t n, double *a, int sign)
{
    for (int64_t i = 1, j = 0; i < n; ++i) {
        int64_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double tr = a[2 * i], ti = a[2 * i + 1];
            a[2 * i] = a[2 * j]; a[2 * i + 1] = a[2 * j + 1];
            a[2 * j] = tr; a[2 * j + 1] = ti;
        }
    }
    for (int64_t len = 2; len <= n; len <<= 1) {
        double ang = sign * 2.0 * M_PI / (double)len;
        for (int64_t i = 0; i < n; i += len)
            for (int64_t k = 0; k < len / 2; ++k) {
                double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                double *p = a + 2 * (i + k), *q = a + 2 * (i + k + len / 2);
                double xr = q[0] * wr - q[1] * wi, xi = q[0] * wi + q[1] * wr;
                q[0] = p[0] - xr; q[1] = p[1] - xi;
                p[0] += xr; p[1] += xi;
            }
    }
}

/* unnormalised DFT of length n with kernel exp(sign*2*pi*i*jk/n), in place */
static int dft1(int64_t n, double *a, int sign)
{
    if ((n & (n - 1)) == 0) { fft_pow2(n, a, sign); return 0; }
    int64_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    double *wch = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *A = (double *)calloc(2 * (size_t)m, sizeof(double));
    double *B = (double *)calloc(2 * (size_t)m, sizeof(double));
    if (!wch || !A || !B) { free(wch); free(A); free(B); return -1; }
    for (int64_t k = 0; k < n; ++k) {
        int64_t kk = (k * k) % (2 * n);
        double ang = sign * M_PI * (double)kk / (double)n;
        wch[2 * k] = cos(ang); wch[2 * k + 1] = sin(ang);
    }
    for (int64_t k = 0; k < n; ++k) {
        A[2 * k] = a[2 * k] * wch[2 * k] - a[2 * k + 1] * wch[2 * k + 1];
        A[2 * k + 1] = a[2 * k] * wch[2 * k + 1] + a[2 * k + 1] * wch[2 * k];
    }
    B[0] = wch[0]; B[1] = -wch[1];
    for (int64_t k = 1; k < n; ++k) {
        B[2 * k] = B[2 * (m - k)] = wch[2 * k];
        B[2 * k + 1] = B[2 * (m - k) + 1] = -wch[2 * k + 1];
    }
    fft_pow2(m, A, -1); fft_pow2(m, B, -1);
    for (int64_t k = 0; k < m; ++k) {
        double r = A[2 * k] * B[2 * k] - A[2 * k + 1] * B[2 * k + 1];
        double i = A[2 * k] * B[2 * k + 1] + A[2 * k + 1] * B[2 * k];
        A[2 * k] = r; A[2 * k + 1] = i;
    }
    fft_pow2(m, A, +1);
    for (int64_t k = 0; k < n; ++k) {
        double r = A[2 * k] / (double)m, i = A[2 * k + 1] / (double)m;
        a[2 * k] = r * wch[2 * k] - i * wch[2 * k + 1];
        a[2 * k + 1] = r * wch[2 * k + 1] + i * wch[2 * k];
    }
    free(wch); free(A); free(B);
    return 0;
}

/* unnormalised 2-D DFT, in place, row-major N x N */
static int dft2(int64_t N, double *a, int sign)
{
    for (int64_t y = 0; y < N; ++y)
        if (dft1(N, a + 2 * y * N, sign)) return -1;
    double *col = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    if (!col) return -1;
    for (int64_t x = 0; x < N; ++x) {
        for (int64_t y = 0; y < N; ++y) { col[2 * y] = a[2 * (y * N + x)]; col[2 * y + 1] = a[2 * (y * N + x) + 1]; }
        if (dft1(N, col, sign)) { free(col); return -1; }
        for (int64_t y = 0; y < N; ++y) { a[2 * (y * N + x)] = col[2 * y]; a[2 * (y * N + x) + 1] = col[2 * y + 1]; }
    }
    free(col);
    return 0;
}

/* roll: out[i] = in[(i + s) mod N] on both axes.
 * shift2D uses s = ceil(N/2) (== numpy fftshift), ishift2D s = floor(N/2) (== ifftshift)
 * [accelerate-fft Data.Array.Accelerate.Math.DFT.Centre, un-vendored]. */
static void roll2(int64_t N, const double *in, double *out, int64_t s)
{
    for (int64_t y = 0; y < N; ++y)
        for (int64_t x = 0; x < N; ++x) {
            int64_t sy = (y + s) % N, sx = (x + s) % N;
            out[2 * (y * N + x)] = in[2 * (sy * N + sx)];
            out[2 * (y * N + x) + 1] = in[2 * (sy * N + sx) + 1];
        }
}

/* fft / ifft of src/Gridding.hs:815-829: shift2D . fft2D mode . ishift2D.
 * accelerate-fft: Forward = exp(-...), unnormalised; Inverse = exp(+...), scaled 1/N^2. */
int gridref_fft2_centered(int64_t N, const double *in, double *out, int inverse)
{
    double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)N * (size_t)N);
    if (!tmp) return -1;
    roll2(N, in, tmp, N / 2);
    if (dft2(N, tmp, inverse ? +1 : -1)) { free(tmp); return -1; }
    if (inverse) {
        double sc = 1.0 / ((double)N * (double)N);
        for (int64_t k = 0; k < 2 * N * N; ++k) tmp[k] *= sc;
    }
    roll2(N, tmp, out, (N + 1) / 2);
    free(tmp);
    return 0;
}

/* ---- pad_mid (:682-691) via padder (:863-877) ---------------------------------------------
 * padder reads `array ! index2 oldx oldy` (:875): the input is TRANSPOSED while it is padded.
 * pad_mid returns ff untouched (no transpose) when n == n0 (:688). */
static void pad_mid(int64_t n0, const double *ff, int64_t n, double *out)
{
    if (n == n0) { memcpy(out, ff, sizeof(double) * 2 * (size_t)n * (size_t)n); return; }
    int64_t p0 = n / 2 - n0 / 2;
    memset(out, 0, sizeof(double) * 2 * (size_t)n * (size_t)n);
    for (int64_t y = 0; y < n; ++y)
        for (int64_t x = 0; x < n; ++x) {
            int64_t oldx = x - p0, oldy = y - p0;
            if (oldx >= 0 && oldx < n0 && oldy >= 0 && oldy < n0) {
                out[2 * (y * n + x)] = ff[2 * (oldx * n0 + oldy)];
                out[2 * (y * n + x) + 1] = ff[2 * (oldx * n0 + oldy) + 1];
            }
        }
}

/* ---- convolve2d, src/Gridding.hs:795-811 -------------------------------------------------
 *   m = 2^ceil(log2(2n-1)); a?fft = fft2D Inverse . ishift2D $ pad_mid a? m
 *   convolved = shift2D . fft2D Forward $ a1fft*a2fft ; extract_mid convolved n ; * m^2 */
void gridref_convolve2d(int64_t n, const double *a1, const double *a2, double *out)
{
    int64_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    size_t sz = sizeof(double) * 2 * (size_t)m * (size_t)m;
    double *p1 = (double *)malloc(sz), *p2 = (double *)malloc(sz), *t = (double *)malloc(sz);
    pad_mid(n, a1, m, t); roll2(m, t, p1, m / 2); dft2(m, p1, +1);
    pad_mid(n, a2, m, t); roll2(m, t, p2, m / 2); dft2(m, p2, +1);
    double sc = 1.0 / ((double)m * (double)m);
    for (int64_t k = 0; k < m * m; ++k) {
        double ar = p1[2 * k] * sc, ai = p1[2 * k + 1] * sc;
        double br = p2[2 * k] * sc, bi = p2[2 * k + 1] * sc;
        t[2 * k] = ar * br - ai * bi;
        t[2 * k + 1] = ar * bi + ai * br;
    }
    dft2(m, t, -1);
    roll2(m, t, p1, (m + 1) / 2);
    /* extract_mid (:694-707): rows/cols [m/2 - n/2, +n) */
    int64_t c = m / 2 - n / 2;
    double m2 = (double)(m * m);
    for (int64_t y = 0; y < n; ++y)
        for (int64_t x = 0; x < n; ++x) {
            out[2 * (y * n + x)] = p1[2 * ((c + y) * m + (c + x))] * m2;
            out[2 * (y * n + x) + 1] = p1[2 * ((c + y) * m + (c + x)) + 1] * m2;
        }
    free(p1); free(p2); free(t);
}

/* direct form: out = same_conv(a1, a2)^T, i.e.
 *   out[x][y] = sum_{i,j} a1[i][j] * a2[y-i+c][x-j+c],  c = n/2 */
void gridref_convolve2d_direct(int64_t n, const double *a1, const double *a2, double *out)
{
    int64_t c = n / 2;
    for (int64_t y = 0; y < n; ++y)
        for (int64_t x = 0; x < n; ++x) {
            double sr = 0.0, si = 0.0;
            for (int64_t i = 0; i < n; ++i) {
                int64_t yy = y - i + c;
                if (yy < 0 || yy >= n) continue;
                for (int64_t j = 0; j < n; ++j) {
                    int64_t xx = x - j + c;
                    if (xx < 0 || xx >= n) continue;
                    double ar = a1[2 * (i * n + j)], ai = a1[2 * (i * n + j) + 1];
                    double br = a2[2 * (yy * n + xx)], bi = a2[2 * (yy * n + xx) + 1];
                    sr += ar * br - ai * bi;
                    si += ar * bi + ai * br;
                }
            }
            out[2 * (x * n + y)] = sr;
            out[2 * (x * n + y) + 1] = si;
        }
}

/* ---- aw_kernel_fn2, src/Gridding.hs:761-775 ---------------------------------------------- */
void gridref_aw_kernel_fn2(int64_t Q, int64_t S, int64_t yf, int64_t xf, const double *wkern,
                           const double *a1, const double *a2, double *out, int direct)
{
    double *ak = (double *)malloc(sizeof(double) * 2 * (size_t)S * (size_t)S);
    const double *wk = wkern + 2 * ((yf * Q + xf) * S * S);
    if (direct) {
        gridref_convolve2d_direct(S, a1, a2, ak);
        gridref_convolve2d_direct(S, ak, wk, out);
    } else {
        gridref_convolve2d(S, a1, a2, ak);
        gridref_convolve2d(S, ak, wk, out);
    }
    free(ak);
}

/* ---- convgrid3 / convgrid4, src/Gridding.hs:246-396 -----------------------------------------
 *   awkern = conj (aw_kernel_fn2 yf xf w[wbin] akerns[a1] akerns[a2])   (:294, :392)
 *   G[y0+i, x0+j] += vis * awkern[i,j], bounds fixed as in convgrid     (:297-317, :361-377) */
void gridref_awgrid(int64_t H, int64_t Wd, double *G, int64_t n, int64_t W, int64_t Q,
                    int64_t S, int64_t A, const double *wkerns, const double *akerns,
                    const double *u, const double *v, const int64_t *wbin,
                    const int64_t *a1, const int64_t *a2, const double *vis, int direct)
{
    (void)W; (void)A;
    double *aw = (double *)malloc(sizeof(double) * 2 * (size_t)S * (size_t)S);
    for (int64_t k = 0; k < n; ++k) {
        int64_t x, xf, y, yf;
        frac_coord1(Wd, Q, u[k], &x, &xf);
        frac_coord1(H, Q, v[k], &y, &yf);
        gridref_aw_kernel_fn2(Q, S, yf, xf, wkerns + 2 * (wbin[k] * Q * Q * S * S),
                              akerns + 2 * (a1[k] * S * S), akerns + 2 * (a2[k] * S * S), aw,
                              direct);
        int64_t x0 = x - S / 2, y0 = y - S / 2;
        double vr = vis[2 * k], vi = vis[2 * k + 1];
        for (int64_t i = 0; i < S; ++i)
            for (int64_t j = 0; j < S; ++j) {
                int64_t xx = x0 + j, yy = y0 + i;
                if (xx < 0 || yy < 0 || xx >= Wd || yy >= H) continue;
                double kr = aw[2 * (i * S + j)], ki = -aw[2 * (i * S + j) + 1];
                G[2 * (yy * Wd + xx)] += vr * kr - vi * ki;
                G[2 * (yy * Wd + xx) + 1] += vr * ki + vi * kr;
            }
    }
    free(aw);
}

/* ---- w_kernel, src/Gridding.hs:610-728 --------------------------------------------------------
 *   coordinates2 n (:637-648): base[k] = (-(n div 2))*(1/n) + k*(1/n); l[y,x]=base[x], m[y,x]=base[y]
 *   kernel_coordinates (:621-635): (l,m) * theta   (shifts / transform matrix default to none)
 *   w_kernel_function (:651-667): exp(i*2*pi*w*(1 - sqrt(1 - l^2 - m^2)))
 *   kernel_oversample (:669-680): pad_mid ff (n*qpx) -> ifft -> extract_oversampled
 *   extract_oversampled (:709-728): K[yf,xf,y,x] = af[c - yf + qpx*y, c - xf + qpx*x] * qpx^2,
 *                                   c = na/2 - qpx*(s/2) */
int gridref_w_kernel(double theta, double w, int64_t npixFF, int64_t npixKern, int64_t qpx,
                     double *out)
{
    int64_t n = npixFF, s = npixKern, na = n * qpx;
    double *ff = (double *)malloc(sizeof(double) * 2 * (size_t)n * (size_t)n);
    double *pad = (double *)malloc(sizeof(double) * 2 * (size_t)na * (size_t)na);
    double *af = (double *)malloc(sizeof(double) * 2 * (size_t)na * (size_t)na);
    if (!ff || !pad || !af) { free(ff); free(pad); free(af); return -1; }
    double step = 1.0 / (double)n;
    double start = (double)(-(n / 2)) * step;
    for (int64_t y = 0; y < n; ++y)
        for (int64_t x = 0; x < n; ++x) {
            double l = (start + (double)x * step) * theta;
            double m = (start + (double)y * step) * theta;
            double r2 = l * l + m * m;
            double ph = 1.0 - sqrt(1.0 - r2);
            double arg = 2.0 * M_PI * w * ph;
            ff[2 * (y * n + x)] = cos(arg);
            ff[2 * (y * n + x) + 1] = sin(arg);
        }
    pad_mid(n, ff, na, pad);
    int rc = gridref_fft2_centered(na, pad, af, 1);
    if (!rc) {
        int64_t c = na / 2 - qpx * (s / 2);
        double q2 = (double)(qpx * qpx);
        for (int64_t yf = 0; yf < qpx; ++yf)
            for (int64_t xf = 0; xf < qpx; ++xf)
                for (int64_t y = 0; y < s; ++y)
                    for (int64_t x = 0; x < s; ++x) {
                        int64_t ny = c - yf + qpx * y, nx = c - xf + qpx * x;
                        double *o = out + 2 * ((((yf * qpx) + xf) * s + y) * s + x);
                        o[0] = af[2 * (ny * na + nx)] * q2;
                        o[1] = af[2 * (ny * na + nx) + 1] * q2;
                    }
    }
    free(ff); free(pad); free(af);
    return rc;
}
