// AW-projection gridders: convgrid3 / convgrid4 (src/Gridding.hs:246-396), which produce the same
// grid, and aw_kernel_fn2 / convolve2d (:761-811) that build each visibility's kernel
//     awkern_k = conj( convolve2d( convolve2d(akerns[a1], akerns[a2]), wkerns[wbin, yf, xf] ) ).
//
// The reference evaluates convolve2d with six 32x32 FFTs per visibility inside a sequential
// `awhile`.  Its pad_mid transposes the operands (padder reads `array ! index2 oldx oldy`, :875),
// so convolve2d(a, b) = same_conv(a, b)^T exactly; here that is evaluated directly in LDS:
//   1. the antenna-pair product convolve2d(a1, a2) is computed once per pair that occurs;
//   2. one work-group per visibility convolves it with its w-kernel slice and conjugates;
//   3. the per-visibility kernels feed the same LDS-tile gridder as convgrid2 (per_vis mode).
// Work is done in batches so the per-visibility kernel buffer stays bounded.
#include "common.h"

namespace gridhip {

// out[x*S + y] = sum_{i,j} a[i][j] * b[y-i+c][x-j+c]   (c = S/2; a, b in LDS)   == same_conv(a,b)^T
__device__ __forceinline__ void conv_same_T(const double2 *a, const double2 *b, int S, double2 *out, bool conj)
{
    const int c = S / 2;
    for (int o = threadIdx.x; o < S * S; o += blockDim.x) {
        const int x = o / S, y = o - x * S;
        double sr = 0.0, si = 0.0;
        const int ilo = max(0, y + c - (S - 1)), ihi = min(S - 1, y + c);
        const int jlo = max(0, x + c - (S - 1)), jhi = min(S - 1, x + c);
        for (int i = ilo; i <= ihi; ++i) {
            const double2 *arow = a + i * S;
            const double2 *brow = b + (y - i + c) * S + (x + c);
            for (int j = jlo; j <= jhi; ++j) {
                const double2 av = arow[j], bv = brow[-j];
                sr += av.x * bv.x - av.y * bv.y;
                si += av.x * bv.y + av.y * bv.x;
            }
        }
        out[o] = make_double2(sr, conj ? -si : si);
    }
}

__global__ void aw_mark_pairs_kernel(int64_t n, int64_t A, const int64_t *__restrict__ a1,
                                     const int64_t *__restrict__ a2, int32_t *__restrict__ flag)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = a1[k], q = a2[k];
        if (p >= 0 && p < A && q >= 0 && q < A) flag[p * A + q] = 1;
    }
}

// exclusive scan of flags -> slot index of every used pair; single work-group
__global__ void __launch_bounds__(1024) aw_scan_pairs_kernel(int64_t m, const int32_t *__restrict__ flag,
                                                             int32_t *__restrict__ slot, int32_t *__restrict__ total)
{
    __shared__ int32_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (m + 1023) / 1024;
    const int64_t lo = tid * per, hi = min(lo + per, m);
    int s = 0;
    for (int64_t i = lo; i < hi; ++i) s += flag[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 1024; ++i) {
            int t = part[i];
            part[i] = acc;
            acc += t;
        }
        *total = acc;
    }
    __syncthreads();
    int acc = part[tid];
    for (int64_t i = lo; i < hi; ++i) {
        slot[i] = flag[i] ? acc : -1;
        acc += flag[i];
    }
}

// akern[slot] = convolve2d(akerns[a1], akerns[a2]) for every used pair (one work-group per pair index)
__global__ void aw_pair_kernel(int64_t A, int S, const double2 *__restrict__ akerns, const int32_t *__restrict__ slot,
                               double2 *__restrict__ pairk)
{
    extern __shared__ double2 sm[];
    const int64_t pq = blockIdx.x;
    const int32_t s = slot[pq];
    if (s < 0) return;
    const int64_t p = pq / A, q = pq - p * A;
    double2 *la = sm, *lb = sm + S * S;
    for (int t = threadIdx.x; t < S * S; t += blockDim.x) {
        la[t] = akerns[p * S * S + t];
        lb[t] = akerns[q * S * S + t];
    }
    __syncthreads();
    conv_same_T(la, lb, S, pairk + (size_t)s * S * S, false);
}

// kperv[k] = conj(convolve2d(pairk[pair_k], wkerns[wbin_k, yf_k, xf_k]))  (one work-group per visibility)
__global__ void aw_vis_kernel(int64_t H, int64_t Wd, int64_t n, int64_t W, int32_t Q, int S, int64_t A,
                              const double2 *__restrict__ wkerns, const double2 *__restrict__ pairk,
                              const int32_t *__restrict__ slot, const double *__restrict__ u,
                              const double *__restrict__ v, int64_t stride, const int64_t *__restrict__ wbin,
                              const int64_t *__restrict__ a1, const int64_t *__restrict__ a2,
                              double2 *__restrict__ kperv, int32_t *__restrict__ scalars)
{
    extern __shared__ double2 sm[];
    const int64_t k = blockIdx.x;
    if (k >= n) return;
    double2 *out = kperv + (size_t)k * S * S;
    const int64_t wb = wbin[k], p = a1[k], q = a2[k];
    const double pu = u[k * stride], pv = v[k * stride];
    const bool bad = wb < 0 || wb >= W || p < 0 || p >= A || q < 0 || q >= A || !(pu == pu) || !(pv == pv);
    if (bad) {  // the reference would index out of range; contribute nothing and count it
        for (int t = threadIdx.x; t < S * S; t += blockDim.x) out[t] = make_double2(0.0, 0.0);
        if (threadIdx.x == 0 && (pu == pu) && (pv == pv)) atomicAdd(&scalars[1], 1);
        return;
    }
    int64_t x, y;
    int32_t xf, yf;
    frac_coord_dev(Wd, Q, pu, &x, &xf);
    frac_coord_dev(H, Q, pv, &y, &yf);
    const double2 *wk = wkerns + ((size_t)(wb * Q + yf) * Q + xf) * S * S;
    const double2 *pk = pairk + (size_t)slot[p * A + q] * S * S;
    double2 *la = sm, *lb = sm + S * S;
    for (int t = threadIdx.x; t < S * S; t += blockDim.x) {
        la[t] = pk[t];
        lb[t] = wk[t];
    }
    __syncthreads();
    conv_same_T(la, lb, S, out, true);
}


// The same per-visibility kernels for a compile-time support, organised for the vector ALU: a wave takes four
// visibilities at a time, lane = (visibility v, column x), and every lane keeps out[y][x] for all S rows y in
// registers.  For each row i of the pair kernel the lane holds a[i][0..S) in registers (fetched one row ahead)
// and, for every y whose w-kernel row r = y - i + c exists, adds sum_j a[i][j] * b[r][x + c - j]; b comes
// from an LDS copy of the w-kernel slice padded with c zero columns on both sides, so the inner loops have
// no bounds logic and skip whole rows only (uniformly).  One LDS read and four FMAs per complex product
// against two reads, the FMAs and ~8 address/loop instructions in the generic kernel above.
template <int S>
__global__ void __launch_bounds__(256) aw_vis_rows_kernel(int64_t H, int64_t Wd, int64_t n, int64_t W, int32_t Q,
                                                          int64_t A, const double2 *__restrict__ wkerns,
                                                          const double2 *__restrict__ pairk,
                                                          const int32_t *__restrict__ slot,
                                                          const double *__restrict__ u, const double *__restrict__ v,
                                                          int64_t stride, const int64_t *__restrict__ wbin,
                                                          const int64_t *__restrict__ a1,
                                                          const int64_t *__restrict__ a2, double2 *__restrict__ kperv,
                                                          int32_t *__restrict__ scalars)
{
    constexpr int C = S / 2, PW = S + 2 * C, S2 = S * S;
    static_assert(S <= 16, "one 16-lane row per visibility");
    extern __shared__ double2 sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v4 = lane >> 4, x = lane & 15;
    double2 *bp = sm + (size_t)(wave * 4 + v4) * S * PW;  // this visibility's padded w-kernel slice
    // pair-kernel rows pass through a two-deep LDS ring (one element per lane and row, fetched one row ahead)
    double2 *aring = sm + (size_t)16 * S * PW + (size_t)wave * 2 * 64;
    const int64_t groups = (n + 3) / 4;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t k = grp * 4 + v4;
        const bool have = k < n;
        int64_t wb = 0, p = 0, q = 0;
        double pu = 0.0, pv = 0.0;
        if (have) {
            wb = wbin[k];
            p = a1[k];
            q = a2[k];
            pu = u[k * stride];
            pv = v[k * stride];
        }
        const bool bad = wb < 0 || wb >= W || p < 0 || p >= A || q < 0 || q >= A || !(pu == pu) || !(pv == pv);
        const bool live = have && !bad;
        int64_t cx = 0, cy = 0;
        int32_t xf = 0, yf = 0;
        if (live) {
            frac_coord_dev(Wd, Q, pu, &cx, &xf);
            frac_coord_dev(H, Q, pv, &cy, &yf);
        }
        const double2 *wk = wkerns + (live ? ((size_t)(wb * Q + yf) * Q + xf) * S2 : 0);
        const double2 *pk = pairk + (live ? (size_t)slot[p * A + q] * S2 : 0);
        // padded copy of the slice: row r, padded column pc holds b[r][pc - C]
        __builtin_amdgcn_wave_barrier();  // (the previous group's reads of this LDS region are done: same wave)
        {   // all of the lane's loads first, then its LDS stores (a load-store loop would pay the memory latency per trip)
            constexpr int NE = (S * PW + 15) / 16;
            double2 stage[NE];
#pragma unroll
            for (int t = 0; t < NE; ++t) {
                const int e = x + 16 * t;
                const int r = e / PW, pc = e - r * PW, cc = pc - C;
                stage[t] = (live && e < S * PW && cc >= 0 && cc < S) ? wk[r * S + cc] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int t = 0; t < NE; ++t) {
                const int e = x + 16 * t;
                if (e < S * PW) bp[e] = stage[t];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // acc[r] belongs to output row y = r + i - C while pair-kernel row i is being applied (it meets w-kernel
        // row r there); after each i the array moves down by one and the row that has just received its last
        // contribution leaves.  So every index below is a compile-time constant.
        double2 acc[S];
#pragma unroll
        for (int r = 0; r < S; ++r) acc[r] = make_double2(0.0, 0.0);
        double2 *out = kperv + (size_t)(have ? k : 0) * S2 + (size_t)x * S;  // out[x * S + y]: comes out transposed
        const bool store = have && x < S;
        double2 arow[S], b0[S], b1[S];  // b0 / b1: w-kernel row r for even / odd r, one fetched ahead
        const int xa = x < S ? x : S - 1;
        double2 anext = pk[xa];  // this lane's element of row 0
        for (int i = 0; i < S; ++i) {
            aring[(i & 1) * 64 + lane] = anext;
            if (i + 1 < S) anext = pk[(i + 1) * S + xa];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < S; ++j) arow[j] = aring[(i & 1) * 64 + v4 * 16 + j];  // a[i][j] of this lane's visibility
            const int rlo = max(0, C - i), rhi = min(S - 1, S - 1 + C - i);  // rows r whose y = r + i - C exists
            {
                const double2 *brow = bp + rlo * PW + (x + 2 * C);  // b[r][x + C - j] = brow[-j]
                if (rlo & 1) {
#pragma unroll
                    for (int j = 0; j < S; ++j) b1[j] = brow[-j];
                } else {
#pragma unroll
                    for (int j = 0; j < S; ++j) b0[j] = brow[-j];
                }
            }
#pragma unroll
            for (int r = 0; r < S; ++r) {
                if (r < rlo || r > rhi) continue;  // uniform
                double2(&bc)[S] = (r & 1) ? b1 : b0;
                double2(&bn)[S] = (r & 1) ? b0 : b1;
                if (r + 1 <= rhi) {
                    const double2 *brow = bp + (r + 1) * PW + (x + 2 * C);
#pragma unroll
                    for (int j = 0; j < S; ++j) bn[j] = brow[-j];
                }
                // four independent chains (one FMA each per product): with one wave per SIMD nothing else hides
                // the FMA latency
                double pr = 0.0, qr = 0.0, pi = 0.0, qi = 0.0;
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    pr = fma(arow[j].x, bc[j].x, pr);
                    qr = fma(arow[j].y, bc[j].y, qr);
                    pi = fma(arow[j].x, bc[j].y, pi);
                    qi = fma(arow[j].y, bc[j].x, qi);
                }
                acc[r].x += pr - qr;
                acc[r].y += pi + qi;
            }
            // acc[0] is row y = i - C: complete when that exists
            if (i >= C && store) out[i - C] = live ? make_double2(acc[0].x, -acc[0].y) : make_double2(0.0, 0.0);
#pragma unroll
            for (int r = 0; r + 1 < S; ++r) acc[r] = acc[r + 1];
            acc[S - 1] = make_double2(0.0, 0.0);
        }
        // after the shift that followed i = S-1, acc[r] is row y = r + S - C
        if (store) {
#pragma unroll
            for (int r = 0; r + S - C < S; ++r)
                out[r + S - C] = live ? make_double2(acc[r].x, -acc[r].y) : make_double2(0.0, 0.0);
            if (bad && x == 0 && (pu == pu) && (pv == pv)) atomicAdd(&scalars[1], 1);
        }
    }
}

}  // namespace gridhip

using namespace gridhip;

extern "C" {

// Device pointers; asynchronous except for scratch allocation and one read-back of the pair count.
int gridhip_awgrid_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W, int64_t Q,
                       int64_t S, int64_t A, const double *wkerns, const double *akerns, const double *u,
                       const double *v, int64_t uv_stride, const int64_t *wbin, const int64_t *a1,
                       const int64_t *a2, const double *vis)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (H <= 0 || Wd <= 0 || n < 0 || W <= 0 || Q <= 0 || S <= 0 || A <= 0 || uv_stride < 1 || !grid || !wkerns ||
        !akerns || (n > 0 && (!u || !v || !wbin || !a1 || !a2 || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    if (S > 63 || A > 46340 || n > (int64_t)0x7fffff00) return fail(ctx, GRIDHIP_EUNSUPPORTED, "shape outside aw limits");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    GH_CHECK_HIP(ctx, hipMemsetAsync(ctx->d_scalars, 0, 16 * sizeof(int32_t), ctx->stream));
    if (n == 0) return GRIDHIP_OK;
    const size_t S2 = (size_t)S * S, pairs = (size_t)A * A;
    const size_t lds = 2 * S2 * sizeof(double2);

    // ---- antenna-pair kernels
    void *dflag = nullptr, *dslot = nullptr, *dpairk = nullptr, *dkperv = nullptr;
    int rc = GRIDHIP_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        if (dflag) (void)hipFree(dflag);
        if (dslot) (void)hipFree(dslot);
        if (dpairk) (void)hipFree(dpairk);
        if (dkperv) (void)hipFree(dkperv);
    };
#define AW_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) {                                                                           \
            rc = fail(ctx, e__ == hipErrorOutOfMemory ? GRIDHIP_ENOMEM : GRIDHIP_EHIP, "%s failed: %s", #call, \
                      hipGetErrorString(e__));                                                             \
            cleanup();                                                                                     \
            return rc;                                                                                     \
        }                                                                                                  \
    } while (0)
    AW_HIP(hipMalloc(&dflag, pairs * 4 + 16));
    AW_HIP(hipMalloc(&dslot, pairs * 4));
    AW_HIP(hipMemsetAsync(dflag, 0, pairs * 4 + 16, ctx->stream));
    int64_t blocks = (n + 255) / 256;
    if (blocks > ctx->num_cu * 8) blocks = ctx->num_cu * 8;
    hipLaunchKernelGGL(aw_mark_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n, A, a1, a2,
                       (int32_t *)dflag);
    int32_t *dtotal = (int32_t *)dflag + pairs;
    hipLaunchKernelGGL(aw_scan_pairs_kernel, dim3(1), dim3(1024), 0, ctx->stream, (int64_t)pairs,
                       (const int32_t *)dflag, (int32_t *)dslot, dtotal);
    int32_t used = 0;
    AW_HIP(hipMemcpyAsync(&used, dtotal, 4, hipMemcpyDeviceToHost, ctx->stream));
    AW_HIP(hipStreamSynchronize(ctx->stream));
    AW_HIP(hipMalloc(&dpairk, (size_t)(used > 0 ? used : 1) * S2 * 16));
    hipLaunchKernelGGL(aw_pair_kernel, dim3((unsigned)pairs), dim3(256), lds, ctx->stream, A, (int)S,
                       (const double2 *)akerns, (const int32_t *)dslot, (double2 *)dpairk);

    // ---- per-visibility kernels + gridding, in batches
    const int64_t batch = n < (1 << 21) ? n : (1 << 21);  // 2M x 3.6 KB = 7.5 GB at 15x15
    AW_HIP(hipMalloc(&dkperv, (size_t)batch * S2 * 16));
    for (int64_t lo = 0; lo < n; lo += batch) {
        const int64_t m = n - lo < batch ? n - lo : batch;
        if (S == 15) {
            constexpr int S_ = 15;
            const size_t rows_lds = ((size_t)16 * S_ * (S_ + 2 * (S_ / 2)) + 4 * 2 * 64) * sizeof(double2);  // 16 padded slices + row rings
            if (!(ctx->attr_mask & 4u)) {
                AW_HIP(hipFuncSetAttribute((const void *)aw_vis_rows_kernel<S_>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds));
                ctx->attr_mask |= 4u;
            }
            int64_t rblocks = (m + 15) / 16;
            if (rblocks > ctx->num_cu) rblocks = ctx->num_cu;
            hipLaunchKernelGGL(aw_vis_rows_kernel<S_>, dim3((unsigned)rblocks), dim3(256), rows_lds, ctx->stream, H, Wd,
                               m, W, (int32_t)Q, A, (const double2 *)wkerns, (const double2 *)dpairk,
                               (const int32_t *)dslot, u + lo * uv_stride, v + lo * uv_stride, uv_stride, wbin + lo,
                               a1 + lo, a2 + lo, (double2 *)dkperv, ctx->d_scalars);
        } else
        hipLaunchKernelGGL(aw_vis_kernel, dim3((unsigned)m), dim3(256), lds, ctx->stream, H, Wd, m, W, (int32_t)Q,
                           (int)S, A, (const double2 *)wkerns, (const double2 *)dpairk, (const int32_t *)dslot,
                           u + lo * uv_stride, v + lo * uv_stride, uv_stride, wbin + lo, a1 + lo, a2 + lo,
                           (double2 *)dkperv, ctx->d_scalars);
        AW_HIP(hipGetLastError());
        rc = grid_per_vis_kernels(ctx, H, Wd, grid, m, Q, S, S, (const double *)dkperv, u + lo * uv_stride,
                                  v + lo * uv_stride, uv_stride, vis + 2 * lo);
        if (rc != GRIDHIP_OK) {
            cleanup();
            return rc;
        }
    }
#undef AW_HIP
    cleanup();
    return GRIDHIP_OK;
}

// Host pointers: convgrid3 / convgrid4 of src/Gridding.hs:246-396 (same result).
int gridhip_awgrid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W, int64_t Q, int64_t S,
                   int64_t A, const double *wkerns, const double *akerns, const double *u, const double *v,
                   int64_t uv_stride, const int64_t *wbin, const int64_t *a1, const int64_t *a2, const double *vis)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (H <= 0 || Wd <= 0 || n < 0 || W <= 0 || Q <= 0 || S <= 0 || A <= 0 || uv_stride < 1 || !grid || !wkerns ||
        !akerns || (n > 0 && (!u || !v || !wbin || !a1 || !a2 || !vis)))
        return fail(ctx, GRIDHIP_EINVAL, "bad argument");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)H * Wd, span = n > 0 ? (size_t)(n - 1) * uv_stride + 1 : 1;
    const size_t wel = (size_t)W * Q * Q * S * S, ael = (size_t)A * S * S;
    struct Buf {
        void *p = nullptr;
        ~Buf()
        {
            if (p) (void)hipFree(p);
        }
    } dg, du, dv, dwb, da1, da2, dvis, dwk, dak;
    GH_CHECK_HIP(ctx, hipMalloc(&dg.p, cells * 16));
    GH_CHECK_HIP(ctx, hipMalloc(&du.p, span * 8));
    GH_CHECK_HIP(ctx, hipMalloc(&dv.p, span * 8));
    GH_CHECK_HIP(ctx, hipMalloc(&dwb.p, (size_t)n * 8 + 8));
    GH_CHECK_HIP(ctx, hipMalloc(&da1.p, (size_t)n * 8 + 8));
    GH_CHECK_HIP(ctx, hipMalloc(&da2.p, (size_t)n * 8 + 8));
    GH_CHECK_HIP(ctx, hipMalloc(&dvis.p, (size_t)n * 16 + 16));
    GH_CHECK_HIP(ctx, hipMalloc(&dwk.p, wel * 16));
    GH_CHECK_HIP(ctx, hipMalloc(&dak.p, ael * 16));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dg.p, grid, cells * 16, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dwk.p, wkerns, wel * 16, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dak.p, akerns, ael * 16, hipMemcpyHostToDevice, ctx->stream));
    if (n > 0) {
        GH_CHECK_HIP(ctx, hipMemcpyAsync(du.p, u, span * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(dv.p, v, span * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(dwb.p, wbin, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(da1.p, a1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(da2.p, a2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(dvis.p, vis, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    }
    GH_CHECK(gridhip_awgrid_dev(ctx, H, Wd, (double *)dg.p, n, W, Q, S, A, (const double *)dwk.p,
                                (const double *)dak.p, (const double *)du.p, (const double *)dv.p, uv_stride,
                                (const int64_t *)dwb.p, (const int64_t *)da1.p, (const int64_t *)da2.p,
                                (const double *)dvis.p));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(grid, dg.p, cells * 16, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

}  // extern "C"
