// Un-binned kernels: `grid` (nearest-cell scatter, src/Gridding.hs:95-112) and the direct
// global-atomic form of convgrid2 (src/Gridding.hs:199-244) kept as the measured baseline the
// LDS-tile design is compared against ("variant" = 1) and as the path for kernels wider than
// 64 columns.
#include "common.h"

namespace gridhip {

__global__ void __launch_bounds__(256) simple_grid_kernel(int64_t H, int64_t Wd, double *__restrict__ grid,
                                                          int64_t n, const double *__restrict__ u,
                                                          const double *__restrict__ v, int64_t stride,
                                                          const double2 *__restrict__ vis)
{
#pragma clang fp contract(off)
    const int64_t halfn = H / 2;
    const double nf = (double)H;  // the reference takes n from the grid height (:101-103)
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const double pu = u[k * stride], pv = v[k * stride];
        if (!(pu == pu) || !(pv == pv)) continue;
        const double fx = floor(0.5 + nf * pu), fy = floor(0.5 + nf * pv);
        if (!(fabs(fx) < 4.0e18) || !(fabs(fy) < 4.0e18)) continue;
        const int64_t x = halfn + (int64_t)fx, y = halfn + (int64_t)fy;
        if (x < 0 || y < 0 || x >= Wd || y >= H) continue;
        const double2 val = vis[k];
        double *dst = grid + 2 * (y * Wd + x);
        unsafeAtomicAdd(dst, val.x);
        unsafeAtomicAdd(dst + 1, val.y);
    }
}

// one wave per visibility, lanes over taps, every tap a pair of global fp64 atomics
__global__ void __launch_bounds__(256) direct_grid_kernel(int64_t H, int64_t Wd, double *__restrict__ grid,
                                                          int64_t n, int32_t W, int32_t Q, int32_t gh, int32_t gw,
                                                          const double2 *__restrict__ gcf,
                                                          const double *__restrict__ u,
                                                          const double *__restrict__ v, int64_t stride,
                                                          const int64_t *__restrict__ wbin,
                                                          const double2 *__restrict__ vis,
                                                          int32_t *__restrict__ scalars)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int S2 = gh * gw;
    for (int64_t k = wave0; k < n; k += nwaves) {
        const double pu = u[k * stride], pv = v[k * stride];
        if (!(pu == pu) || !(pv == pv)) continue;
        int64_t x, y;
        int32_t xf, yf;
        frac_coord_dev(Wd, Q, pu, &x, &xf);
        frac_coord_dev(H, Q, pv, &y, &yf);
        const int64_t x0 = x - gw / 2, y0 = y - gh / 2;
        if (x0 <= -(int64_t)gw || x0 >= Wd || y0 <= -(int64_t)gh || y0 >= H) continue;
        const int64_t wb = wbin ? wbin[k] : 0;
        if (wb < 0 || wb >= W) {
            if (lane == 0) atomicAdd(&scalars[0], 1);
            continue;
        }
        const double2 val = vis[k];
        const double2 *kp = gcf + ((size_t)(wb * Q + yf) * Q + xf) * S2;
        for (int t = lane; t < S2; t += 64) {
            const int i = t / gw, j = t - i * gw;
            const int64_t xx = x0 + j, yy = y0 + i;
            if (xx < 0 || yy < 0 || xx >= Wd || yy >= H) continue;
            const double2 kv = kp[t];
            double *dst = grid + 2 * (yy * Wd + xx);
            unsafeAtomicAdd(dst, val.x * kv.x - val.y * kv.y);
            unsafeAtomicAdd(dst + 1, val.x * kv.y + val.y * kv.x);
        }
    }
}

int launch_simple_grid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, const double *u,
                       const double *v, int64_t uv_stride, const double *vis)
{
    if (n <= 0) return GRIDHIP_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > ctx->num_cu * 8) blocks = ctx->num_cu * 8;
    hipLaunchKernelGGL(simple_grid_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, H, Wd, grid, n, u, v,
                       uv_stride, (const double2 *)vis);
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

int launch_direct_grid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W, int64_t Q,
                       int64_t gh, int64_t gw, const double *gcf, const double *u, const double *v,
                       int64_t uv_stride, const int64_t *wbin, const double *vis)
{
    GH_CHECK_HIP(ctx, hipMemsetAsync(ctx->d_scalars, 0, 16 * sizeof(int32_t), ctx->stream));
    if (n <= 0) return GRIDHIP_OK;
    int64_t blocks = (n + 3) / 4;  // 4 waves per block, one visibility per wave step
    if (blocks > ctx->num_cu * 8) blocks = ctx->num_cu * 8;
    hipLaunchKernelGGL(direct_grid_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, H, Wd, grid, n,
                       (int32_t)W, (int32_t)Q, (int32_t)gh, (int32_t)gw, (const double2 *)gcf, u, v, uv_stride,
                       wbin, (const double2 *)vis, ctx->d_scalars);
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

}  // namespace gridhip
