// C-ABI entry points of the gridders (include/gridhip.h): argument checks, the device-pointer
// forms that enqueue the kernels, and the host-pointer drop-in forms that stage through HBM.

#include "common.h"

using namespace gridhip;

namespace {

int check_common(gridhip_ctx *ctx, int64_t H, int64_t Wd, const void *grid, int64_t n, const void *u,
                 const void *v, int64_t uv_stride)
{
    if (!ctx) return GRIDHIP_EINVAL;
    if (H <= 0 || Wd <= 0 || n < 0 || uv_stride < 1)
        return fail(ctx, GRIDHIP_EINVAL, "bad size (H=%lld Wd=%lld n=%lld uv_stride=%lld)", (long long)H,
                    (long long)Wd, (long long)n, (long long)uv_stride);
    if (!grid || (n > 0 && (!u || !v))) return fail(ctx, GRIDHIP_EINVAL, "null pointer");
    if (n > (int64_t)0x7fffff00) return fail(ctx, GRIDHIP_EUNSUPPORTED, "n must be < 2^31 per call");
    return GRIDHIP_OK;
}

// bump allocator over the staging workspace
struct Stage {
    char *base;
    size_t off = 0;
    template <typename T>
    T *take(size_t count)
    {
        T *p = reinterpret_cast<T *>(base + off);
        off += (count * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

size_t aligned(size_t b) { return (b + 255) & ~(size_t)255; }

// the host-pointer forms are synchronous: an internal consistency failure of the call (a record that did not fit
// the record array, a slice index outside the kernel table - the caller's arrays changed during the call, or a
// bug) is reported instead of a silently incomplete grid
int check_errors(gridhip_ctx *ctx)
{
    int32_t e = 0;
    GH_CHECK_HIP(ctx, hipMemcpyAsync(&e, ctx->d_scalars + 2, sizeof e, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (e) return fail(ctx, GRIDHIP_EINVAL, "internal consistency check failed for %d records (inputs modified during the call?)", e);
    return GRIDHIP_OK;
}


// kernel table -> zero-padded square parts: out[(slice * P + part)][sub][sub] (Geom, "Sub-footprints")
__global__ void __launch_bounds__(256) pad_kernels_kernel(Geom g, int64_t nslices, const double2 *__restrict__ in,
                                                          double2 *__restrict__ out)
{
    const int sub = g.gh, S2 = sub * sub;
    const int64_t total = nslices * g.P * S2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e % S2);
        const int64_t sp = e / S2;
        const int part = (int)(sp % g.P);
        const int64_t slice = sp / g.P;
        const int i = (part / g.px) * sub + t / sub, j = (part % g.px) * sub + t % sub;
        out[e] = (i < g.fgh && j < g.fgw) ? in[(slice * g.fgh + i) * g.fgw + j] : make_double2(0.0, 0.0);
    }
}
}  // namespace

namespace gridhip {

// How a kernel shape the tap-reusing tile kernel has no instantiation for is cut into square parts it has one for:
// py x px parts of side sub = max(ceil(gh / py), ceil(gw / px)), the fewest parts with sub <= side (32: the largest
// square the tap-reusing kernel is instantiated for; 16 under option "subfoot", round 2's cut).
static bool choose_parts(int64_t gh, int64_t gw, int side, int *py, int *px, int *sub)
{
    const int y = (int)((gh + side - 1) / side), x = (int)((gw + side - 1) / side);
    const int sy = (int)((gh + y - 1) / y), sx = (int)((gw + x - 1) / x);
    int sb = sy > sx ? sy : sx;
    if (sb < 5) sb = 5;  // (smallest instantiation)
    if (y * x > 64) return false;
    *py = y;
    *px = x;
    *sub = sb;
    return true;
}

// make_geom sizes the work-group for the general tile kernel's LDS (two planes); the tap-reusing kernel keeps its im
// plane at a fixed distance from the re plane, so small tiles take more LDS there and fewer work-groups share a CU:
// give each the waves that leaves free (2048^2, 7 x 7, 10^6 visibilities: one work-group per CU, 16 waves instead of
// 8 - 0.227 -> 0.194 ms per call)
static void sorted_block(gridhip_ctx *ctx, Prep *p)
{
    if (ctx->opt.block != 0) return;
    const size_t per_cu = (size_t)ctx->max_lds / p->lds_sorted;
    p->block = per_cu >= 4 ? 256 : per_cu >= 2 ? 512 : 1024;
}

int prepare(gridhip_ctx *ctx, int64_t H, int64_t Wd, int64_t W, int64_t Q, int64_t gh, int64_t gw, int64_t n, Prep *p)
{
    *p = Prep();
    p->nrec = n;
    int rc = make_geom(ctx, H, Wd, W, Q, gh, gw, n, &p->g, &p->block, &p->lds);
    if (rc == GRIDHIP_OK) {
        // sort: 0 = auto (on when a work item holds enough visibilities for slices to repeat), 1 = on, 2 = off
        const bool want = ctx->opt.sort == 1 || (ctx->opt.sort == 0 && n / (int64_t)p->g.nbins >= 256);
        // option "subfoot" = 1: supports above 16 go through sub-footprints (one record per spatial part) as in round 2,
        // instead of the tile kernel's parts of the tap list (one record per visibility) - kept for comparison runs
        const bool old_parts = ctx->opt.subfoot == 1 && gh > 16;
        p->sorted = want && !old_parts && sorted_plan(ctx, p->g, p->block, &p->nkeys, &p->batch, &p->lds_sorted);
        // a sorted work item may span several LDS batches; keep it big enough to flush each tile once
        if (p->sorted) {
            p->g.chunk = p->batch;
            sorted_block(ctx, p);
        }
        if (p->sorted || ctx->opt.sort == 2) return GRIDHIP_OK;
    } else if (rc != GRIDHIP_EUNSUPPORTED || ctx->opt.tile != 0)
        return rc;
    // sub-footprints: supports above 16 and non-square kernels as P records of a small square support each
    int py, px, sub;
    const int side = ctx->opt.subfoot == 1 ? 16 : 32;
    if (ctx->opt.sort != 2 && (gh != gw || gh > side || gh < 5) && choose_parts(gh, gw, side, &py, &px, &sub) &&
        n * (int64_t)(py * px) < (int64_t)0x7fffff00 && W * Q * Q * (int64_t)(py * px) < ((int64_t)1 << 30)) {
        Prep q;
        const int P = py * px;
        q.nrec = n * P;
        if (make_geom(ctx, H, Wd, W, Q, sub, sub, q.nrec, &q.g, &q.block, &q.lds, P) == GRIDHIP_OK) {
            q.g.fgh = (int32_t)gh;
            q.g.fgw = (int32_t)gw;
            q.g.py = py;
            q.g.px = px;
            q.g.P = P;
            q.g.nvis = (int32_t)(n > 0 ? n : 1);  // (nrec stays n * P)
            q.g.nslices = (int32_t)(W * Q * Q * P);
            set_rec_bits(&q.g);
            set_div_magic(&q.g);
            const bool want = ctx->opt.sort == 1 || (ctx->opt.sort == 0 && q.nrec / (int64_t)q.g.nbins >= 256);
            q.sorted = want && sorted_plan(ctx, q.g, q.block, &q.nkeys, &q.batch, &q.lds_sorted);
            if (q.sorted) {
                q.g.chunk = q.batch;
                sorted_block(ctx, &q);
                *p = q;
                return GRIDHIP_OK;
            }
        }
    }
    if (rc == GRIDHIP_EUNSUPPORTED) {  // support too large for an LDS tile: direct global-atomic scatter
        p->direct = true;
        return GRIDHIP_OK;
    }
    return rc;
}

int tile_kernels(gridhip_ctx *ctx, const Prep &p, const double *gcf, const double **out)
{
    ctx->last_path = p.direct ? 4 : !p.sorted ? 3 : p.g.P > 1 || p.g.fgh != p.g.gh || p.g.fgw != p.g.gw ? 2 : 1;
    ctx->last_geom[0] = p.g.ngroups;
    ctx->last_geom[1] = p.g.Tx;
    ctx->last_geom[2] = p.g.Ty;
    ctx->last_geom[3] = p.g.imoff > 0;
    *out = gcf;
    if (p.g.P == 1 && p.g.fgh == p.g.gh && p.g.fgw == p.g.gw) return GRIDHIP_OK;
    const int64_t nsl = (int64_t)p.g.W * p.g.Q * p.g.Q;
    const size_t elems = (size_t)nsl * p.g.P * p.g.gh * p.g.gw;
    GH_CHECK(ws_reserve(ctx, ctx->ktab, elems * 16));
    int blocks = (int)((elems + 255) / 256);
    if (blocks > ctx->num_cu * 16) blocks = ctx->num_cu * 16;
    hipLaunchKernelGGL(pad_kernels_kernel, dim3(blocks), dim3(256), 0, ctx->stream, p.g, nsl, (const double2 *)gcf,
                       (double2 *)ctx->ktab.ptr);
    GH_CHECK_HIP(ctx, hipGetLastError());
    *out = (const double *)ctx->ktab.ptr;
    return GRIDHIP_OK;
}

}  // namespace gridhip

extern "C" {

int gridhip_grid_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, const double *u,
                     const double *v, int64_t uv_stride, const double *vis)
{
    GH_CHECK(check_common(ctx, H, Wd, grid, n, u, v, uv_stride));
    if (n > 0 && !vis) return fail(ctx, GRIDHIP_EINVAL, "null vis");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    mark(ctx, 0);
    mark(ctx, 1);
    GH_CHECK(launch_simple_grid(ctx, H, Wd, grid, n, u, v, uv_stride, vis));
    mark(ctx, 2);
    return GRIDHIP_OK;
}

// A record is one 64-bit word: 14 bits of footprint origin, the kernel slice, the visibility's index.  A call whose
// slices x visibilities exceed 2^50 (no realistic one does: 10^8 visibilities leave room for 8 x 10^6 slices) is
// gridded in several parts - gridding and degridding are both sums / maps over visibilities.  Returns the
// visibilities per part, 0 when the call fits, < 0 when no part size does.  Option "rec_bits" (test hook) lowers the
// word's width so that small cases take this path.  (gridhip_last_dropped and "errors" then report the last part's.)
static int64_t part_size(gridhip_ctx *ctx, const Prep &p)
{
    const int limit = ctx->opt.rec_bits >= 16 && ctx->opt.rec_bits < 64 ? (int)ctx->opt.rec_bits : 64;
    if (p.direct || rec_fits(p.g, limit)) return 0;
    const int room = limit - 14 - p.g.kb;
    return room >= 1 ? (int64_t)1 << room : -1;
}

int gridhip_convgrid2_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W, int64_t Q,
                          int64_t gh, int64_t gw, const double *gcf, const double *u, const double *v,
                          int64_t uv_stride, const int64_t *wbin, const double *vis)
{
    GH_CHECK(check_common(ctx, H, Wd, grid, n, u, v, uv_stride));
    if (!gcf || (n > 0 && !vis)) return fail(ctx, GRIDHIP_EINVAL, "null pointer");
    if (W <= 0 || Q <= 0 || gh <= 0 || gw <= 0) return fail(ctx, GRIDHIP_EINVAL, "bad kernel shape");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    Prep p;
    if (ctx->opt.variant == 1)
        p.direct = true;
    else
        GH_CHECK(prepare(ctx, H, Wd, W, Q, gh, gw, n, &p));
    if (const int64_t part = part_size(ctx, p)) {
        if (part < 0) return fail(ctx, GRIDHIP_EUNSUPPORTED, "kernel table with too many slices");
        for (int64_t lo = 0; lo < n; lo += part)
            GH_CHECK(gridhip_convgrid2_dev(ctx, H, Wd, grid, part < n - lo ? part : n - lo, W, Q, gh, gw, gcf, u + lo * uv_stride,
                                           v + lo * uv_stride, uv_stride, wbin ? wbin + lo : nullptr, vis + 2 * lo));
        return GRIDHIP_OK;
    }
    if (p.direct) {
        ctx->last_path = 4;
        mark(ctx, 0);
        mark(ctx, 1);
        GH_CHECK(launch_direct_grid(ctx, H, Wd, grid, n, W, Q, gh, gw, gcf, u, v, uv_stride, wbin, vis));
        mark(ctx, 2);
        return GRIDHIP_OK;
    }
    // scratch is sized before the timed region begins
    GH_CHECK(ws_reserve(ctx, ctx->tables, tables_bytes(p.g)));
    GH_CHECK(ws_reserve(ctx, ctx->recs, (size_t)(p.nrec > 0 ? p.nrec : 1) * sizeof(RecWord)));
    mark(ctx, 0);
    const double *tk = gcf;
    GH_CHECK(tile_kernels(ctx, p, gcf, &tk));
    GH_CHECK(launch_bin(ctx, p.g, p.nrec, u, v, uv_stride, wbin));
    mark(ctx, 1);
    if (n > 0) {
        if (p.sorted)
            GH_CHECK(launch_tile_grid_sorted(ctx, p.g, p.block, p.lds_sorted, p.nkeys, p.batch, p.nrec, tk, vis, grid, false));
        else
            GH_CHECK(launch_tile_grid(ctx, p.g, p.block, p.lds, p.nrec, tk, vis, grid));
    }
    mark(ctx, 2);
    return GRIDHIP_OK;
}

int gridhip_convgrid_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t Q, int64_t gh,
                         int64_t gw, const double *gcf, const double *u, const double *v, int64_t uv_stride,
                         const double *vis)
{
    // convgrid is convgrid2 with a single plane and wbin = 0 (src/Gridding.hs:196 vs :243)
    return gridhip_convgrid2_dev(ctx, H, Wd, grid, n, 1, Q, gh, gw, gcf, u, v, uv_stride, nullptr, vis);
}

int gridhip_degrid2_dev(gridhip_ctx *ctx, int64_t H, int64_t Wd, const double *grid, int64_t n, int64_t W,
                        int64_t Q, int64_t gh, int64_t gw, const double *gcf, const double *u, const double *v,
                        int64_t uv_stride, const int64_t *wbin, double *vis_out)
{
    GH_CHECK(check_common(ctx, H, Wd, grid, n, u, v, uv_stride));
    if (!gcf || (n > 0 && !vis_out)) return fail(ctx, GRIDHIP_EINVAL, "null pointer");
    if (W <= 0 || Q <= 0 || gh <= 0 || gw <= 0) return fail(ctx, GRIDHIP_EINVAL, "bad kernel shape");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    Prep p;
    GH_CHECK(prepare(ctx, H, Wd, W, Q, gh, gw, n, &p));
    if (p.direct) return fail(ctx, GRIDHIP_EUNSUPPORTED, "degrid2: support %lldx%lld too large for an LDS tile", (long long)gh, (long long)gw);
    if (const int64_t part = part_size(ctx, p)) {
        if (part < 0) return fail(ctx, GRIDHIP_EUNSUPPORTED, "kernel table with too many slices");
        for (int64_t lo = 0; lo < n; lo += part)
            GH_CHECK(gridhip_degrid2_dev(ctx, H, Wd, grid, part < n - lo ? part : n - lo, W, Q, gh, gw, gcf, u + lo * uv_stride,
                                         v + lo * uv_stride, uv_stride, wbin ? wbin + lo : nullptr, vis_out + 2 * lo));
        return GRIDHIP_OK;
    }
    GH_CHECK(ws_reserve(ctx, ctx->tables, tables_bytes(p.g)));
    GH_CHECK(ws_reserve(ctx, ctx->recs, (size_t)(p.nrec > 0 ? p.nrec : 1) * sizeof(RecWord)));
    mark(ctx, 0);
    // visibilities with no tap inside the grid (or an out-of-range wbin) predict 0: the counting sweep writes those
    // zeros.  Sub-footprints, and supports above 16 x 16 in the tap-reusing kernel (parts of the tap list), sum a
    // visibility's parts with atomics, so there the whole array starts from zero.
    const bool parts = p.g.P > 1 || (p.sorted && p.g.gh > 16) || ctx->opt.fault_inject > 0;  // (the test hook loses records: clear as well)
    if (n > 0 && parts) GH_CHECK_HIP(ctx, hipMemsetAsync(vis_out, 0, (size_t)n * 16, ctx->stream));
    const double *tk = gcf;
    GH_CHECK(tile_kernels(ctx, p, gcf, &tk));
    GH_CHECK(launch_bin(ctx, p.g, p.nrec, u, v, uv_stride, wbin, parts ? nullptr : reinterpret_cast<double2 *>(vis_out)));
    mark(ctx, 1);
    if (n > 0) {
        if (p.sorted)  // the sorted kernel's degrid mode reads `grid` and writes the vis array
            GH_CHECK(launch_tile_grid_sorted(ctx, p.g, p.block, p.lds_sorted, p.nkeys, p.batch, p.nrec, tk, vis_out,
                                             const_cast<double *>(grid), true));
        else
            GH_CHECK(launch_tile_degrid(ctx, p.g, p.block, p.lds, p.nrec, tk, grid, vis_out));
    }
    mark(ctx, 2);
    return GRIDHIP_OK;
}

}  // extern "C"

extern "C" {

// ---------------------------------------------------------------------------------------------
// host-pointer forms

static int stage_uv(gridhip_ctx *ctx, Stage &st, int64_t n, int64_t stride, const double *u, const double *v,
                    double **du, double **dv)
{
    const size_t span = n > 0 ? (size_t)(n - 1) * stride + 1 : 0;
    *du = st.take<double>(span ? span : 1);
    *dv = st.take<double>(span ? span : 1);
    if (span) {
        GH_CHECK_HIP(ctx, hipMemcpyAsync(*du, u, span * 8, hipMemcpyHostToDevice, ctx->stream));
        GH_CHECK_HIP(ctx, hipMemcpyAsync(*dv, v, span * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    return GRIDHIP_OK;
}

int gridhip_grid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, const double *u,
                 const double *v, int64_t uv_stride, const double *vis)
{
    GH_CHECK(check_common(ctx, H, Wd, grid, n, u, v, uv_stride));
    if (n > 0 && !vis) return fail(ctx, GRIDHIP_EINVAL, "null vis");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)H * Wd, span = n > 0 ? (size_t)(n - 1) * uv_stride + 1 : 1;
    GH_CHECK(ws_reserve(ctx, ctx->stage, aligned(cells * 16) + 2 * aligned(span * 8) + aligned((size_t)n * 16 + 16)));
    Stage st{(char *)ctx->stage.ptr};
    double *dg = st.take<double>(cells * 2), *du, *dv;
    GH_CHECK(stage_uv(ctx, st, n, uv_stride, u, v, &du, &dv));
    double *dvis = st.take<double>((size_t)n * 2 + 2);
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dg, grid, cells * 16, hipMemcpyHostToDevice, ctx->stream));
    if (n) GH_CHECK_HIP(ctx, hipMemcpyAsync(dvis, vis, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK(gridhip_grid_dev(ctx, H, Wd, dg, n, du, dv, uv_stride, dvis));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(grid, dg, cells * 16, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

int gridhip_convgrid2(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t W, int64_t Q,
                      int64_t gh, int64_t gw, const double *gcf, const double *u, const double *v,
                      int64_t uv_stride, const int64_t *wbin, const double *vis)
{
    GH_CHECK(check_common(ctx, H, Wd, grid, n, u, v, uv_stride));
    if (!gcf || (n > 0 && !vis)) return fail(ctx, GRIDHIP_EINVAL, "null pointer");
    if (W <= 0 || Q <= 0 || gh <= 0 || gw <= 0) return fail(ctx, GRIDHIP_EINVAL, "bad kernel shape");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)H * Wd, span = n > 0 ? (size_t)(n - 1) * uv_stride + 1 : 1;
    const size_t kel = (size_t)W * Q * Q * gh * gw;
    GH_CHECK(ws_reserve(ctx, ctx->stage, aligned(cells * 16) + 2 * aligned(span * 8) + aligned((size_t)n * 16 + 16) +
                                             aligned((size_t)n * 8 + 8) + aligned(kel * 16)));
    Stage st{(char *)ctx->stage.ptr};
    double *dg = st.take<double>(cells * 2), *du, *dv;
    GH_CHECK(stage_uv(ctx, st, n, uv_stride, u, v, &du, &dv));
    double *dvis = st.take<double>((size_t)n * 2 + 2);
    int64_t *dwb = st.take<int64_t>((size_t)n + 1);
    double *dk = st.take<double>(kel * 2);
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dg, grid, cells * 16, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dk, gcf, kel * 16, hipMemcpyHostToDevice, ctx->stream));
    if (n) GH_CHECK_HIP(ctx, hipMemcpyAsync(dvis, vis, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    if (n && wbin) GH_CHECK_HIP(ctx, hipMemcpyAsync(dwb, wbin, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK(gridhip_convgrid2_dev(ctx, H, Wd, dg, n, W, Q, gh, gw, dk, du, dv, uv_stride, wbin ? dwb : nullptr, dvis));
    if (ctx->opt.variant != 1) GH_CHECK(check_errors(ctx));  // (before the grid is handed back)
    GH_CHECK_HIP(ctx, hipMemcpyAsync(grid, dg, cells * 16, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

int gridhip_convgrid(gridhip_ctx *ctx, int64_t H, int64_t Wd, double *grid, int64_t n, int64_t Q, int64_t gh,
                     int64_t gw, const double *gcf, const double *u, const double *v, int64_t uv_stride,
                     const double *vis)
{
    return gridhip_convgrid2(ctx, H, Wd, grid, n, 1, Q, gh, gw, gcf, u, v, uv_stride, nullptr, vis);
}

int gridhip_degrid2(gridhip_ctx *ctx, int64_t H, int64_t Wd, const double *grid, int64_t n, int64_t W, int64_t Q,
                    int64_t gh, int64_t gw, const double *gcf, const double *u, const double *v, int64_t uv_stride,
                    const int64_t *wbin, double *vis_out)
{
    GH_CHECK(check_common(ctx, H, Wd, grid, n, u, v, uv_stride));
    if (!gcf || (n > 0 && !vis_out)) return fail(ctx, GRIDHIP_EINVAL, "null pointer");
    if (W <= 0 || Q <= 0 || gh <= 0 || gw <= 0) return fail(ctx, GRIDHIP_EINVAL, "bad kernel shape");
    GH_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)H * Wd, span = n > 0 ? (size_t)(n - 1) * uv_stride + 1 : 1;
    const size_t kel = (size_t)W * Q * Q * gh * gw;
    GH_CHECK(ws_reserve(ctx, ctx->stage, aligned(cells * 16) + 2 * aligned(span * 8) + aligned((size_t)n * 16 + 16) +
                                             aligned((size_t)n * 8 + 8) + aligned(kel * 16)));
    Stage st{(char *)ctx->stage.ptr};
    double *dg = st.take<double>(cells * 2), *du, *dv;
    GH_CHECK(stage_uv(ctx, st, n, uv_stride, u, v, &du, &dv));
    double *dvis = st.take<double>((size_t)n * 2 + 2);
    int64_t *dwb = st.take<int64_t>((size_t)n + 1);
    double *dk = st.take<double>(kel * 2);
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dg, grid, cells * 16, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK_HIP(ctx, hipMemcpyAsync(dk, gcf, kel * 16, hipMemcpyHostToDevice, ctx->stream));
    if (n && wbin) GH_CHECK_HIP(ctx, hipMemcpyAsync(dwb, wbin, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK(gridhip_degrid2_dev(ctx, H, Wd, dg, n, W, Q, gh, gw, dk, du, dv, uv_stride, wbin ? dwb : nullptr, dvis));
    GH_CHECK(check_errors(ctx));
    if (n) GH_CHECK_HIP(ctx, hipMemcpyAsync(vis_out, dvis, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GRIDHIP_OK;
}

}  // extern "C"
