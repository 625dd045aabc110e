// Tile kernels: the w-projection gridder (convgrid / convgrid2, src/Gridding.hs:153-244) and
// its gather twin (degrid2) on CDNA4.
//
// One work item = (w-group, grid tile, chunk of <=chunk binned visibilities).  A work-group
// keeps the tile plus its kernel-support halo — (T+gh-1) x (T+gw-1) complex cells, planar
// re/im, 100 KB at T=64, 15x15 — in LDS (160 KB per CU).  Each wave takes one visibility at a
// time: its 32-byte record arrives by scalar load, lanes map to kernel taps (rw lanes per
// kernel row, 64/rw rows per step) so the tap read from the [gh][gw] slice is one coalesced
// run and the accumulate is a conflict-free ds_add_f64 (row pitch chosen in ctx.hip).  The
// grid read-modify-write of the reference's `permute (+)` therefore never leaves the CU; HBM
// sees the tile once, when the work-group flushes it with global_atomic_add_f64 (neighbouring
// tiles overlap in their halos, and several chunks/groups may share a tile).
//
// fixoutofbounds (:883-891): the LDS region may hang over the grid edge; cells outside the
// grid are simply not flushed, i.e. out-of-range taps are dropped, never wrapped.
#include "common.h"

namespace gridhip {

struct WorkItem {
    int tile, v_lo, v_hi;
};

// Map blockIdx -> work item.  Work items of w-group g are the blocks with blockIdx % ngroups
// == g, so with the dispatcher's round-robin over the 8 XCDs a group's kernel planes stay in
// one XCD's L2 (speed only; any placement is correct).
__device__ __forceinline__ bool find_work(const Geom &g, const int32_t *__restrict__ bin_start,
                                          const int32_t *__restrict__ work_start, WorkItem *w)
{
    const int grp = blockIdx.x % g.ngroups;
    const int k = blockIdx.x / g.ngroups;
    const int32_t *ws = work_start + (size_t)grp * (g.ntiles + 1);
    if (k >= ws[g.ntiles]) return false;
    int lo = 0, hi = g.ntiles;  // largest t with ws[t] <= k
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ws[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    const int bin = grp * g.ntiles + lo;
    const int c = k - ws[lo], nch = ws[lo + 1] - ws[lo];
    const int b0 = bin_start[bin], cnt = bin_start[bin + 1] - b0;
    w->tile = lo;
    w->v_lo = b0 + (int)(((int64_t)cnt * c) / nch);
    w->v_hi = b0 + (int)(((int64_t)cnt * (c + 1)) / nch);
    return true;
}

__device__ __forceinline__ VisRec load_rec(const VisRec *__restrict__ recs, int idx)
{
    // idx is wave-uniform: let the compiler use scalar loads
    const int4 *p = reinterpret_cast<const int4 *>(recs + idx);
    int4 a = p[0], b = p[1];
    VisRec r;
    r.lxy = a.x;
    r.kslice = a.y;
    r.vr = __hiloint2double(a.w, a.z);
    r.vi = __hiloint2double(b.y, b.x);
    r.orig = b.z;
    r.pad = 0;
    return r;
}

template <int RW>
__global__ void __launch_bounds__(1024) tile_grid_kernel(Geom g, const VisRec *__restrict__ recs,
                                                         const int32_t *__restrict__ bin_start,
                                                         const int32_t *__restrict__ work_start,
                                                         const double2 *__restrict__ gcf,
                                                         double *__restrict__ grid)
{
    extern __shared__ double lds[];
    WorkItem w;
    if (!find_work(g, bin_start, work_start, &w)) return;

    const int tid = threadIdx.x;
    const int plane = g.lrows * g.ldw;  // doubles per plane
    {
        double2 *z = reinterpret_cast<double2 *>(lds);
        for (int i = tid; i < plane; i += blockDim.x) z[i] = make_double2(0.0, 0.0);  // 2*plane doubles
    }
    __syncthreads();

    constexpr int RPI = 64 / RW;  // kernel rows per wave step
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int j = lane % RW, ri = lane / RW;
    const bool jok = j < g.gw;
    const int S2 = g.gh * g.gw;
    double *lre = lds, *lim = lds + plane;

    for (int vi = w.v_lo + wave; vi < w.v_hi; vi += nw) {
        const VisRec r = load_rec(recs, __builtin_amdgcn_readfirstlane(vi));
        const double2 *kp = gcf + (size_t)r.kslice * S2 + j;
        const int lbase = (r.lxy >> 16) * g.ldw + (r.lxy & 0xffff) + j;
        for (int i = ri; i < g.gh; i += RPI) {
            if (jok) {
                const double2 kv = kp[i * g.gw];
                const double re = r.vr * kv.x - r.vi * kv.y;
                const double im = r.vr * kv.y + r.vi * kv.x;
                const int a = lbase + i * g.ldw;
                __hip_atomic_fetch_add(&lre[a], re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&lim[a], im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    // flush the cells that exist in the grid
    const int tx = w.tile % g.ntx, ty = w.tile / g.ntx;
    const int64_t ox = (int64_t)tx * g.T - g.offx, oy = (int64_t)ty * g.T - g.offy;
    const int ncell = g.lrows * g.lcols;
    for (int c = tid; c < ncell; c += blockDim.x) {
        const int r_ = c / g.lcols, c_ = c - r_ * g.lcols;
        const int64_t gx = ox + c_, gy = oy + r_;
        if (gx < 0 || gy < 0 || gx >= g.Wd || gy >= g.H) continue;
        const double re = lre[r_ * g.ldw + c_], im = lim[r_ * g.ldw + c_];
        if (re == 0.0 && im == 0.0) continue;
        double *dst = grid + 2 * (gy * g.Wd + gx);
        unsafeAtomicAdd(dst, re);
        unsafeAtomicAdd(dst + 1, im);
    }
}

// Gather twin: vis_out[orig] = sum_ij gcf[kslice][i][j] * G[y0+i][x0+j]; the tile (zero outside
// the grid) is staged in LDS once per work item, taps are summed across the wave.
template <int RW>
__global__ void __launch_bounds__(1024) tile_degrid_kernel(Geom g, const VisRec *__restrict__ recs,
                                                           const int32_t *__restrict__ bin_start,
                                                           const int32_t *__restrict__ work_start,
                                                           const double2 *__restrict__ gcf,
                                                           const double2 *__restrict__ grid,
                                                           double2 *__restrict__ vis_out)
{
    extern __shared__ double lds[];
    WorkItem w;
    if (!find_work(g, bin_start, work_start, &w)) return;

    const int tid = threadIdx.x;
    const int plane = g.lrows * g.ldw;
    double *lre = lds, *lim = lds + plane;
    const int tx = w.tile % g.ntx, ty = w.tile / g.ntx;
    const int64_t ox = (int64_t)tx * g.T - g.offx, oy = (int64_t)ty * g.T - g.offy;
    const int ncell = g.lrows * g.lcols;
    for (int c = tid; c < ncell; c += blockDim.x) {
        const int r_ = c / g.lcols, c_ = c - r_ * g.lcols;
        const int64_t gx = ox + c_, gy = oy + r_;
        double2 v = make_double2(0.0, 0.0);
        if (gx >= 0 && gy >= 0 && gx < g.Wd && gy < g.H) v = grid[gy * g.Wd + gx];
        lre[r_ * g.ldw + c_] = v.x;
        lim[r_ * g.ldw + c_] = v.y;
    }
    __syncthreads();

    constexpr int RPI = 64 / RW;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int j = lane % RW, ri = lane / RW;
    const bool jok = j < g.gw;
    const int S2 = g.gh * g.gw;

    for (int vi = w.v_lo + wave; vi < w.v_hi; vi += nw) {
        const VisRec r = load_rec(recs, __builtin_amdgcn_readfirstlane(vi));
        const double2 *kp = gcf + (size_t)r.kslice * S2 + j;
        const int lbase = (r.lxy >> 16) * g.ldw + (r.lxy & 0xffff) + j;
        double sr = 0.0, si = 0.0;
        for (int i = ri; i < g.gh; i += RPI) {
            if (jok) {
                const double2 kv = kp[i * g.gw];
                const int a = lbase + i * g.ldw;
                const double gr = lre[a], gi = lim[a];
                sr += kv.x * gr - kv.y * gi;
                si += kv.x * gi + kv.y * gr;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sr += __shfl_xor(sr, off, 64);
            si += __shfl_xor(si, off, 64);
        }
        if (lane == 0) vis_out[r.orig] = make_double2(sr, si);
    }
}

template <typename K>
static int raise_lds(gridhip_ctx *ctx, K kernel, uint32_t bit)
{
    if (ctx->attr_mask & bit) return GRIDHIP_OK;
    GH_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          ctx->max_lds));
    ctx->attr_mask |= bit;
    return GRIDHIP_OK;
}

static int work_blocks(const Geom &g, int64_t n)
{
    // upper bound on work items of any one group: every tile may add one partial chunk
    int64_t per_group = n / g.chunk + g.ntiles + 1;
    return (int)(per_group * g.ngroups);
}

int launch_tile_grid(gridhip_ctx *ctx, const Geom &g, int block, size_t lds_bytes, int64_t n,
                     const double *gcf, double *grid)
{
    Tables t = tables_of(ctx, g);
    const VisRec *recs = (const VisRec *)ctx->recs.ptr;
    const dim3 gr(work_blocks(g, n)), bl(block);
#define GH_LAUNCH(RW_, BIT_)                                                                              \
    case RW_:                                                                                             \
        GH_CHECK(raise_lds(ctx, tile_grid_kernel<RW_>, BIT_));                                            \
        hipLaunchKernelGGL(tile_grid_kernel<RW_>, gr, bl, lds_bytes, ctx->stream, g, recs, t.bin_start,   \
                           t.work_start, (const double2 *)gcf, grid);                                     \
        break;
    switch (g.rw) {
        GH_LAUNCH(2, 1u << 4)
        GH_LAUNCH(4, 1u << 5)
        GH_LAUNCH(8, 1u << 6)
        GH_LAUNCH(16, 1u << 7)
        GH_LAUNCH(32, 1u << 8)
        GH_LAUNCH(64, 1u << 9)
        default: return fail(ctx, GRIDHIP_EUNSUPPORTED, "kernel width %d", g.gw);
    }
#undef GH_LAUNCH
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

int launch_tile_degrid(gridhip_ctx *ctx, const Geom &g, int block, size_t lds_bytes, int64_t n,
                       const double *gcf, const double *grid, double *vis_out)
{
    Tables t = tables_of(ctx, g);
    const VisRec *recs = (const VisRec *)ctx->recs.ptr;
    const dim3 gr(work_blocks(g, n)), bl(block);
#define GH_LAUNCH(RW_, BIT_)                                                                               \
    case RW_:                                                                                              \
        GH_CHECK(raise_lds(ctx, tile_degrid_kernel<RW_>, BIT_));                                           \
        hipLaunchKernelGGL(tile_degrid_kernel<RW_>, gr, bl, lds_bytes, ctx->stream, g, recs, t.bin_start,  \
                           t.work_start, (const double2 *)gcf, (const double2 *)grid, (double2 *)vis_out); \
        break;
    switch (g.rw) {
        GH_LAUNCH(2, 1u << 10)
        GH_LAUNCH(4, 1u << 11)
        GH_LAUNCH(8, 1u << 12)
        GH_LAUNCH(16, 1u << 13)
        GH_LAUNCH(32, 1u << 14)
        GH_LAUNCH(64, 1u << 15)
        default: return fail(ctx, GRIDHIP_EUNSUPPORTED, "kernel width %d", g.gw);
    }
#undef GH_LAUNCH
    GH_CHECK_HIP(ctx, hipGetLastError());
    return GRIDHIP_OK;
}

}  // namespace gridhip
